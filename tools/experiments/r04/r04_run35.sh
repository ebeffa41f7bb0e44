#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 55 python -m pytest tests/test_gpu_effdet.py -q -x -k "end_to_end or onnx_file" 2>&1 | tail -2
