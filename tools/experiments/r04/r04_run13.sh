#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for a in "" "--no-overlap" "--no-graph" "--no-graph --no-overlap"; do
  timeout 300 python bench.py --preset c5 --micro-batch 1 --no-cpu-baseline --no-extras --steps 200 --repeats 2 $a > /tmp/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('c5 b1 [$a]:', d['value'], d['ms_per_step'], d['repeats']['fps_median'], d['step_latency_ms']['p50'])"
done
