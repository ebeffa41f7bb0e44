#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest "tests/test_gpu_chain.py::test_micro_batched_step_matches_oracle_chain_fp32" tests/test_gpu_pipeline.py tests/test_gpu_post.py -m gpu -q > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log ); tail -3 $out/pytest.log
for p in c5 c4; do python bench.py --preset $p --no-extras --no-cpu-baseline --repeats 3 > $out/b_$p.json 2>/dev/null; python -c "
import json;d=json.load(open('$out/b_$p.json'));print('$p',d['value'],d['repeats']['fps_median'],d['stages'])"; done
