#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 70 python tools/demo_headless.py --det-type efficientdet --frames 4 2>&1 | tail -6 | cut -c1-200
