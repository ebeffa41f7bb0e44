#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04l
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/profile_layers.py yolov8l --batch 1 --precision fp16 --top 200 > $out/layers_yolov8l_b1_fp16.txt 2>&1
head -1 $out/layers_yolov8l_b1_fp16.txt
awk 'NR>1 {k=$NF; sub(/<.*/,"",k); t[k]+=$1; n[k]++} END {for (k in t) printf "   %-28s %7.3f ms %3d\n", k, t[k], n[k]}' $out/layers_yolov8l_b1_fp16.txt | sort -k2 -n -r | head -6
python tools/profile_layers.py ufldv2_res18 --batch 1 --precision fp16 --top 30 > $out/layers_ufldv2_res18_b1_fp16.txt 2>&1; head -8 $out/layers_ufldv2_res18_b1_fp16.txt | cut -c1-140
( timeout 300 python bench.py --preset c5 --micro-batch 1 --no-cpu-baseline --no-extras --steps 200 --repeats 2 > $out/bench_c5_b1.json 2>/dev/null ); python -c "
import json; d=json.load(open('$out/bench_c5_b1.json')); print('c5 frame-at-a-time:', d['value'], d['ms_per_step'], d['stages'], d['roofline']['all_conv_frac'])"
( timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py tests/test_gpu_configs.py -q -x > $out/pytest_sub.log 2>&1 ); tail -3 $out/pytest_sub.log
