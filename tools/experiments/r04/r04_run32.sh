#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 170 python -m pytest tests/test_gpu_chain.py -q -x 2>&1 | tail -3
