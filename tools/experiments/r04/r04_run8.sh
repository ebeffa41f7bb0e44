#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {
  tag=$1; shift
  args=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --repeats 3 $args > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$out/bench_$tag.json"))
    print("$tag:", d["value"], d["ms_per_step"], d["repeats"]["fps_median"], {k: d["stages"][k] for k in ("det_net_ms","lane_net_ms")})
except Exception as e:
    print("$tag: failed", e, open("$out/bench_$tag.err").read()[-300:])
PY
}
run eager_base "--no-graph" A=1
run eager_l192_d64 "--no-graph" ADAS_CUMASK_LANE=0:191 ADAS_CUMASK_DET=192:255
run eager_l160_d96 "--no-graph" ADAS_CUMASK_LANE=0:159 ADAS_CUMASK_DET=160:255
run eager_l176_d80 "--no-graph" ADAS_CUMASK_LANE=0:175 ADAS_CUMASK_DET=176:255
run eager_l208_d48 "--no-graph" ADAS_CUMASK_LANE=0:207 ADAS_CUMASK_DET=208:255
run graph_l192_d64 "" ADAS_CUMASK_LANE=0:191 ADAS_CUMASK_DET=192:255
run eager_l256_d64 "--no-graph" ADAS_CUMASK_DET=192:255
run graph_base "" A=1
