#!/bin/bash
mkdir -p gpurun_out/r03v7
timeout 400 python -m pytest tests/test_gpu_v7.py -q -m gpu -s > gpurun_out/r03v7/pytest_v7.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03v7/pytest_v7.log
tail -6 gpurun_out/r03v7/pytest_v7.log
ADAS_BENCH_NO_PMC=1 timeout 240 python bench.py --preset v7 --steps 20 --warmup 5 > gpurun_out/r03v7/bench_v7.json 2> gpurun_out/r03v7/bench_v7.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03v7/bench_v7.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stages"], d["repeats"])
e=d["parity"]["e2e"]; print({k:e[k] for k in e if k.startswith("frac")})
PY
