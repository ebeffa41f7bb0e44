#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 150 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_chain.py -q -x -k "equals_components or sink or fp16_matches or yolov8n" 2>&1 | tail -3
