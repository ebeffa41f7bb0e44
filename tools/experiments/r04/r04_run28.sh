#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_pipeline.py -q -k "micro or sink or equals_components" 2>&1 | tail -3
