#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log )
tail -4 $out/pytest_gpu.log
