#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04p
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_effdet.py -q -x -s > $out/pytest_effdet.log 2>&1; echo "exit $?" >> $out/pytest_effdet.log ); tail -40 $out/pytest_effdet.log | cut -c1-250
