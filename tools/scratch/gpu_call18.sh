#!/bin/bash
mkdir -p gpurun_out/r03v7
timeout 400 python -m pytest tests/test_gpu_v7.py tests/test_gpu_nets.py tests/test_gpu_post.py -q -m gpu -s -k "v7 or v5 or leaky" > gpurun_out/r03v7/pytest_v7.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03v7/pytest_v7.log
tail -6 gpurun_out/r03v7/pytest_v7.log
timeout 120 python tools/profile_layers.py yolov7-tiny --batch 64 --precision fp16 --top 12 > gpurun_out/r03v7/layers_v7_b64.txt 2>&1
head -8 gpurun_out/r03v7/layers_v7_b64.txt
timeout 120 python tools/profile_layers.py yolov5n --batch 64 --precision fp16 --top 6 > gpurun_out/r03v7/layers_v5n_b64.txt 2>&1
head -5 gpurun_out/r03v7/layers_v5n_b64.txt
