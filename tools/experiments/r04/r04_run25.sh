#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04y
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_effdet.py -q -x > $out/pytest_se.log 2>&1; echo "exit $?" >> $out/pytest_se.log ); tail -4 $out/pytest_se.log | cut -c1-250
python tools/profile_layers.py efficientdet-d0 --batch 64 --precision fp16 --top 300 > $out/layers_efficientdet-d0_b64_fp16.txt 2>&1; head -1 $out/layers_efficientdet-d0_b64_fp16.txt
awk 'NR>1 {k=$NF; sub(/<.*/,"",k); t[k]+=$1; n[k]++} END {for (k in t) printf "   %-28s %7.3f ms %3d\n", k, t[k], n[k]}' $out/layers_efficientdet-d0_b64_fp16.txt | sort -k2 -n -r | head -8
python tools/profile_layers.py efficientdet-d0 --batch 64 --precision fp16x3 --top 300 > $out/layers_efficientdet-d0_b64_fp16x3.txt 2>&1; head -1 $out/layers_efficientdet-d0_b64_fp16x3.txt
