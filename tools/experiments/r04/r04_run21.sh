#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/sink_debug3.py 2>&1 | tail -12
