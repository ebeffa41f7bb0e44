#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04s
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_configs.py -q -x -k "micro" > $out/pytest_mb.log 2>&1; echo "exit $?" >> $out/pytest_mb.log ); tail -8 $out/pytest_mb.log | cut -c1-250
( ADAS_BENCH_NO_PMC=1 timeout 400 python bench.py --preset c4 --no-cpu-baseline --no-extras --steps 20 --repeats 2 > $out/bench_c4.json 2>$out/bench_c4.err ); python -c "
import json; d=json.load(open('$out/bench_c4.json')); print('c4:', d['value'], d['ms_per_step'], d.get('stages'))"
