#!/bin/bash
mkdir -p gpurun_out/r03v7
timeout 500 python -m pytest tests/test_gpu_v7.py tests/test_gpu_v9.py tests/test_gpu_conv.py -q -m gpu -s -k "v7 or v9s or no_generic_fallback or leaky" > gpurun_out/r03v7/pytest_v7b.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03v7/pytest_v7b.log
tail -5 gpurun_out/r03v7/pytest_v7b.log
timeout 120 python tools/profile_layers.py yolov7-tiny --batch 64 --precision fp16 --top 70 > gpurun_out/r03v7/layers_yolov7-tiny_b64_fp16.txt 2>&1
head -6 gpurun_out/r03v7/layers_yolov7-tiny_b64_fp16.txt
timeout 120 python tools/profile_layers.py yolov9s --batch 64 --precision fp16 --top 12 > gpurun_out/r03v7/layers_yolov9s_b64_fp16.txt 2>&1
head -6 gpurun_out/r03v7/layers_yolov9s_b64_fp16.txt
