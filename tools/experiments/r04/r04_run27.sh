#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_effdet.py tests/test_gpu_onnx_lower.py -q 2>&1 | tail -3
