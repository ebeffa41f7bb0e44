#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log ); tail -8 $out/pytest_gpu.log
( timeout 600 python -m pytest tests/test_gpu_chain.py -q -s -k "fp16 and not x3" > $out/pytest_chain_fp16.log 2>&1 ); grep -E "device fp16|emulated fp16|fp16 S=" $out/pytest_chain_fp16.log | cut -c1-700
( timeout 900 python bench.py --steps 20 > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err ); tail -2 $out/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f/bench.json"))
print(d["value"], d["ms_per_step"], d["config"].get("exact_mode_frames_per_s"))
print(d["config"].get("e2e_vs_fp32_oracle_chain")); print(d["config"].get("exact_mode_e2e_vs_fp32_oracle_chain"))
print(d["modes"]); print(d["stages"])
PY
