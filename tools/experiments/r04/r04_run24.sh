#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04x
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_onnx_lower.py -q -x -s > $out/pytest_lower.log 2>&1; echo "exit $?" >> $out/pytest_lower.log ); grep "lowered\|passed\|failed\|exit" $out/pytest_lower.log | cut -c1-200
