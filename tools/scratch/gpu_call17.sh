#!/bin/bash
mkdir -p gpurun_out/r03v7
timeout 300 python -m pytest tests/test_gpu_v7.py -q -m gpu -s > gpurun_out/r03v7/pytest_v7.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03v7/pytest_v7.log
tail -4 gpurun_out/r03v7/pytest_v7.log
timeout 120 python tools/profile_layers.py yolov7-tiny --batch 64 --precision fp16 --top 70 > gpurun_out/r03v7/layers_v7_b64.txt 2>&1
head -3 gpurun_out/r03v7/layers_v7_b64.txt
