cd $GRAFT_REPO_ROOT
echo "== default, with sync"; timeout 600 python tools/scratch/flaky_pipe.py 100 2>&1 | tail -4
