#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04t
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_v7.py tests/test_gpu_v6.py -q -x > $out/pytest_sink5.log 2>&1; echo "exit $?" >> $out/pytest_sink5.log ); tail -6 $out/pytest_sink5.log | cut -c1-250
