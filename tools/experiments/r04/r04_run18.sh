#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04r
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_chain.py tests/test_gpu_effdet.py -q -x > $out/pytest_sink.log 2>&1; echo "exit $?" >> $out/pytest_sink.log ); tail -6 $out/pytest_sink.log | cut -c1-250
for env in 0 1; do
  ( ADAS_NO_DETECT_SINK=$env ADAS_BENCH_NO_PMC=1 timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 40 --repeats 3 > $out/bench_sink_off$env.json 2>$out/bench_sink_off$env.err ); python -c "
import json; d=json.load(open('$out/bench_sink_off$env.json')); print('ADAS_NO_DETECT_SINK=$env:', d['value'], d['ms_per_step'], d.get('stages'), d['repeats']['fps_median'])"
done
