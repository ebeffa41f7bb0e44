#!/bin/bash
# usage (GPU box, via gpurun): tools/r05_ml_check.sh <tag>  -- the multi-layer launch: parity tests, then an A/B of the bench step with / without it
tag=${1:-r05a}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ml.py -x -q -s > $out/pytest_ml.log 2>&1; echo "pytest exit $?" >> $out/pytest_ml.log ); tail -25 $out/pytest_ml.log
for ml in 0 1; do
  ( ADAS_ML=$ml ADAS_BENCH_NO_PMC=1 timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 30 --repeats 3 > $out/bench_noml$ml.json 2> $out/bench_noml$ml.err; echo "exit $?" >> $out/bench_noml$ml.err )
  python - <<PY
import json
try:
    d=json.load(open('$out/bench_noml$ml.json'))
    print('ADAS_ML=$ml', d['value'], d['ms_per_step'], d['config'].get('stages'), d['config']['kernel_launches_per_step_nets'], d['roofline']['all_conv_frac'], d.get('repeats'))
    for k in d['roofline']['top_kernels']: print('   ', k)
except Exception as ex:
    print('bench failed', ex); print(open('$out/bench_noml$ml.err').read()[-1500:])
PY
done
