#!/bin/bash
mkdir -p gpurun_out/r03s
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py tests/test_gpu_v7.py -q -m gpu -x > gpurun_out/r03s/pytest_subset.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03s/pytest_subset.log
tail -3 gpurun_out/r03s/pytest_subset.log
for m in yolov8n yolov7-tiny yolov8l; do
  b=64; [ $m = yolov8l ] && b=16
  timeout 120 python tools/profile_layers.py $m --batch $b --precision fp16 --top 70 > gpurun_out/r03s/layers_${m}_b${b}_fp16.txt 2>&1
  head -1 gpurun_out/r03s/layers_${m}_b${b}_fp16.txt
done
ADAS_BENCH_NO_PMC=1 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03s/bench_ns.json 2> gpurun_out/r03s/bench_ns.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03s/bench_ns.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stages"], d["repeats"])
PY
