#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03i
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for r in 2 3 4 6 8 10 12 16 8; do ADAS_PRE_ROWS=$r python tools/bench_pre.py 64 2>&1 | tail -1 | tee -a $out/pre_rows.txt; done
