#!/bin/bash
# usage (GPU box): tools/x3_probe2.sh <tag> -- correctness of the shared-weight-tile stream + A/B of it and of the strip-width policy
tag=${1:-x3probe2}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_chain.py -m gpu -q -x > $out/pytest_x3.log 2>&1 ); tail -3 $out/pytest_x3.log
( timeout 900 python tools/ab_bench.py --rounds 2 --steps 30 --repeats 2 --variant base --variant share0:ADAS_H8X_SHARE=0 --variant plan0:ADAS_H8X_PLAN=0 \
    --variant share0plan0:ADAS_H8X_SHARE=0,ADAS_H8X_PLAN=0 --variant narrow:ADAS_LIB=$GRAFT_REPO_ROOT/vehicle-cv-adas_amd/_scratch/libadas_hip_narrow.so > $out/ab_share_plan.txt 2>&1 ); tail -8 $out/ab_share_plan.txt
( timeout 600 python tools/ab_bench.py --rounds 2 --steps 40 --repeats 2 --extra "--precision fp16" --variant base --variant h8plan1:ADAS_H8_PLAN=1 > $out/ab_fp16_plan.txt 2>&1 ); tail -5 $out/ab_fp16_plan.txt
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 40 > $out/layers_ufldv2_res18_b64_fp16x3.txt 2>&1
python tools/profile_layers.py yolov8n --batch 64 --precision fp16x3 --top 100 > $out/layers_yolov8n_b64_fp16x3.txt 2>&1
head -20 $out/layers_ufldv2_res18_b64_fp16x3.txt | cut -c1-150
