#!/bin/bash
out=gpurun_out/r03d; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_post.py tests/test_gpu_v6.py tests/test_gpu_v7.py tests/test_gpu_frontend.py -q -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
for p in v7 v6; do
  ( ADAS_BENCH_NO_PMC=1 timeout 200 python bench.py --preset $p --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err; echo "bench exit $?" >> $out/bench_$p.err )
  python - <<PY
import json
d=json.loads(open('$out/bench_$p.json').read().strip().splitlines()[-1])
print('$p', d["value"], d["ms_per_step"], d["stages"], d["repeats"]["fps_median"], [ (r["kernel"], r.get("us"), r.get("tb_s")) for r in d["post_hbm"][:2]])
PY
done
