#!/bin/bash
mkdir -p gpurun_out/r03v6
timeout 500 python -m pytest tests/test_gpu_v6.py tests/test_gpu_conv.py -q -m gpu -s -k "v6 or transposed or (no_generic_fallback and yolov6)" > gpurun_out/r03v6/pytest_v6.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03v6/pytest_v6.log
tail -6 gpurun_out/r03v6/pytest_v6.log
timeout 120 python tools/profile_layers.py yolov6n --batch 64 --precision fp16 --top 80 > gpurun_out/r03v6/layers_yolov6n_b64_fp16.txt 2>&1
head -12 gpurun_out/r03v6/layers_yolov6n_b64_fp16.txt
ADAS_BENCH_NO_PMC=1 timeout 240 python bench.py --preset v6 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03v6/bench_v6.json 2> gpurun_out/r03v6/bench_v6.err
echo "bench rc $?"; tail -3 gpurun_out/r03v6/bench_v6.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03v6/bench_v6.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stages"], d["repeats"])
e=d["parity"]["e2e"]; print({k:e[k] for k in e if k.startswith("frac")})
PY
