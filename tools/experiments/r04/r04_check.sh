#!/bin/bash
commit=${1:-unknown}
out=$GRAFT_REPO_ROOT/gpurun_out/r04u
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "commit $commit" > $out/commit.txt
( timeout 1200 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log ); tail -4 $out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke exit $?" >> $out/smoke.log ); tail -2 $out/smoke.log
( timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err ); cut -c1-300 $out/bench.json; tail -1 $out/bench.err
