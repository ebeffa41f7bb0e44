#!/bin/bash
# every preset in the split precision: throughput + end-to-end parity of the timed mode against the fp32 oracle chain
out=$GRAFT_REPO_ROOT/gpurun_out/r04q
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for p in c4 c5 v10 v9 v7 v6; do
  ( ADAS_BENCH_NO_PMC=1 timeout 600 python bench.py --preset $p --precision fp16x3 --no-cpu-baseline --steps 10 --repeats 1 > $out/bench_${p}_fp16x3.json 2> $out/bench_${p}_fp16x3.err; echo "exit $?" >> $out/bench_${p}_fp16x3.err )
  python - <<PY
import json
try:
    d=json.load(open('$out/bench_${p}_fp16x3.json')); e=d['parity']['e2e']
    print('$p fp16x3', d['value'], d['ms_per_step'], {k: e.get(k) for k in ('frames','identical_candidate_sets','identical_survivors','identical_track_ids','track_states_compared','lanes_within_1px','max_lane_jump_px')})
except Exception as ex:
    print('$p failed', ex); print(open('$out/bench_${p}_fp16x3.err').read()[-600:])
PY
done
