#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04zz
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -x > $out/pytest_pw.log 2>&1; echo "exit $?" >> $out/pytest_pw.log ); tail -3 $out/pytest_pw.log | cut -c1-200
for env in 0 1; do
  ADAS_NO_PW_WIDE=$env python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_v8n_$env.txt 2>&1
  echo "NO_PW_WIDE=$env: $(head -1 $out/layers_v8n_$env.txt)  conv_pw sum: $(grep conv_pw_kernel $out/layers_v8n_$env.txt | awk '{s+=$1} END {print s}')"
done
for env in 0 1; do
  ( ADAS_NO_PW_WIDE=$env ADAS_BENCH_NO_PMC=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --repeats 3 > $out/bench_$env.json 2>/dev/null ); python -c "
import json; d=json.load(open('$out/bench_$env.json')); print('NO_PW_WIDE=$env:', d['value'], d['repeats']['fps_median'], d['stages']['det_net_ms'])"
done
