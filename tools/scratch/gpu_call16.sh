#!/bin/bash
# YOLOv7-tiny on the GPU: its tests, then the v7 preset of the bench
mkdir -p gpurun_out/r03v7
timeout 300 python -m pytest tests/test_gpu_v7.py -x -q -m gpu -s > gpurun_out/r03v7/pytest_v7.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03v7/pytest_v7.log
tail -5 gpurun_out/r03v7/pytest_v7.log
ADAS_BENCH_NO_PMC=1 timeout 240 python bench.py --preset v7 --steps 20 --warmup 5 > gpurun_out/r03v7/bench_v7.json 2> gpurun_out/r03v7/bench_v7.err
echo "bench rc $?"
tail -c 1500 gpurun_out/r03v7/bench_v7.json
tail -5 gpurun_out/r03v7/bench_v7.err
