#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03l
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 320 512 768 1100 0 320 512 768 1100; do ADAS_HALO_BM128=$v python bench.py --no-extras --no-cpu-baseline --repeats 3 > $out/b_$v.json 2>/dev/null; python -c "
import json;d=json.load(open('$out/b_$v.json'));print('thr=$v',d['value'],d['repeats']['fps_median'],d['stages']['det_net_ms'],d['stages']['lane_net_ms'])" | tee -a $out/sweep.txt; done
