#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_configs.py "tests/test_gpu_conv.py::test_fp16_stores_saturate_instead_of_overflowing" -m gpu -q -s > $out/pytest_chain.log 2>&1; echo "exit $?" >> $out/pytest_chain.log )
grep -v "^$" $out/pytest_chain.log | grep -v "amdgpu.ids" | tail -60
( timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err )
tail -2 $out/bench.err; python -c "
import json;d=json.load(open('$out/bench.json'));print(d['value'],d['modes'],d['parity']['e2e'])"
for p in c4 c5; do
( timeout 900 python bench.py --preset $p --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err; echo "bench exit $?" >> $out/bench_$p.err )
tail -2 $out/bench_$p.err; cat $out/bench_$p.json
done
