#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 75 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_post.py -q -x 2>&1 | tail -3
