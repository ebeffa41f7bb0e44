#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04n
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_x3.py -q -x -k "pointwise or linear or conv_layers or yolov8 or ufldv2 or yolov10" > $out/pytest_x3.log 2>&1; echo "exit $?" >> $out/pytest_x3.log ); tail -5 $out/pytest_x3.log
( timeout 300 python -m pytest tests/test_gpu_conv.py -q -x -k "fc" > $out/pytest_fc.log 2>&1 ); tail -2 $out/pytest_fc.log
python tools/profile_layers.py yolov8n --batch 64 --precision fp16x3 --top 100 > $out/layers_yolov8n_b64_fp16x3.txt 2>&1; head -1 $out/layers_yolov8n_b64_fp16x3.txt
grep "pwx3\|upsample" $out/layers_yolov8n_b64_fp16x3.txt | cut -c1-150
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 100 > $out/layers_ufldv2_res18_b64_fp16x3.txt 2>&1; head -1 $out/layers_ufldv2_res18_b64_fp16x3.txt
grep "cls\|downsample\|pool" $out/layers_ufldv2_res18_b64_fp16x3.txt | cut -c1-150
ADAS_FCX3_TN=2 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 100 2>&1 | grep "cls" | cut -c1-150
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 100 2>&1 | grep "cls" | cut -c1-150
( timeout 400 python bench.py --precision fp16x3 --no-cpu-baseline --no-extras --steps 20 --repeats 2 > $out/bench_x3.json 2>$out/bench_x3.err ); python -c "
import json; d=json.load(open('$out/bench_x3.json')); print('fp16x3:', d['value'], d['ms_per_step'], d.get('stages'))"
