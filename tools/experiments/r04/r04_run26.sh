#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04z
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py tests/test_gpu_configs.py -q -x > $out/pytest_bn80.log 2>&1; echo "exit $?" >> $out/pytest_bn80.log ); tail -4 $out/pytest_bn80.log | cut -c1-250
python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_yolov8n_b64_fp16.txt 2>&1; head -1 $out/layers_yolov8n_b64_fp16.txt; grep "cv3" $out/layers_yolov8n_b64_fp16.txt | cut -c1-150
( ADAS_BENCH_NO_PMC=1 timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 40 --repeats 3 > $out/bench.json 2>$out/bench.err ); python -c "
import json; d=json.load(open('$out/bench.json')); print('north-star:', d['value'], d['ms_per_step'], d.get('stages'), d['repeats']['fps_median'])"
