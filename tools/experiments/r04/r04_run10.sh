#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04j
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/profile_layers.py yolov8l --batch 1 --precision fp16 --top 120 > $out/layers_yolov8l_b1_fp16.txt 2>&1
python tools/profile_layers.py ufldv2_res18 --batch 1 --precision fp16 --top 40 > $out/layers_ufldv2_res18_b1_fp16.txt 2>&1
head -40 $out/layers_yolov8l_b1_fp16.txt | cut -c1-150
head -12 $out/layers_ufldv2_res18_b1_fp16.txt | cut -c1-150
( timeout 300 python bench.py --preset c5 --micro-batch 1 --no-cpu-baseline --no-extras --steps 100 --repeats 2 > $out/bench_c5_b1.json 2>/dev/null ); python -c "
import json; d=json.load(open('$out/bench_c5_b1.json')); print(d['value'], d['ms_per_step'], d['stages'], d['step_latency_ms'])"
