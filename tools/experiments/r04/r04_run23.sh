#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04w
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_v10.py -q -x -k "strip or depthwise" > $out/pytest_strip.log 2>&1; echo "exit $?" >> $out/pytest_strip.log ); tail -5 $out/pytest_strip.log | cut -c1-250
( ADAS_BENCH_NO_PMC=1 timeout 400 python bench.py --preset v10 --no-cpu-baseline --no-extras --steps 30 --repeats 3 > $out/bench_v10.json 2>$out/bench_v10.err ); python -c "
import json; d=json.load(open('$out/bench_v10.json')); print('v10:', d['value'], d['ms_per_step'], d.get('stages'), d['repeats']['fps_median'])"
