#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_gpu_effdet.py tests/test_gpu_v10.py tests/test_gpu_configs.py tests/test_gpu_v9.py tests/test_gpu_v6.py tests/test_gpu_v7.py -q 2>&1 | tail -3
