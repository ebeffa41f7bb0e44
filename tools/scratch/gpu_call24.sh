#!/bin/bash
out=gpurun_out/r03c; mkdir -p $out
for p in c5 c4 v10; do
  ( ADAS_BENCH_NO_PMC=1 timeout 300 python bench.py --preset $p --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err; echo "bench exit $?" >> $out/bench_$p.err )
  tail -1 $out/bench_$p.err
  python - <<PY
import json
d=json.loads(open('$out/bench_$p.json').read().strip().splitlines()[-1])
print('$p', d["value"], d["ms_per_step"], d["stages"], d["repeats"]["fps_median"], d["roofline"]["all_conv_frac"], (d.get("frame_at_a_time") or {}).get("value"))
PY
done
