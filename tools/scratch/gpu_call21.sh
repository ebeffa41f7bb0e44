#!/bin/bash
mkdir -p gpurun_out/r03v7
timeout 600 python -m pytest tests/test_gpu_v7.py tests/test_gpu_v9.py tests/test_gpu_v10.py tests/test_gpu_conv.py -q -m gpu -s -k "v5_layout or yolov9s_c or yolov10s or no_generic_fallback" > gpurun_out/r03v7/pytest_v7c.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03v7/pytest_v7c.log
tail -5 gpurun_out/r03v7/pytest_v7c.log
timeout 120 python tools/profile_layers.py yolov7-tiny --batch 64 --precision fp16 --top 70 > gpurun_out/r03v7/layers_yolov7-tiny_b64_fp16.txt 2>&1
head -3 gpurun_out/r03v7/layers_yolov7-tiny_b64_fp16.txt
timeout 120 python tools/profile_layers.py yolov9c --batch 16 --precision fp16 --top 12 > gpurun_out/r03v7/layers_yolov9c_b16_fp16.txt 2>&1
head -8 gpurun_out/r03v7/layers_yolov9c_b16_fp16.txt
timeout 120 python tools/profile_layers.py yolov10s --batch 64 --precision fp16 --top 12 > gpurun_out/r03v7/layers_yolov10s_b64_fp16.txt 2>&1
head -8 gpurun_out/r03v7/layers_yolov10s_b64_fp16.txt
