cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -q -k "persistent_kernels" -s 2>&1 | tail -8
