cd $GRAFT_REPO_ROOT
for m in 1 2 3; do ADAS_HALO8=$m timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "dma_fed" 2>&1 | tail -3; done
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py tests/test_gpu_engine.py -q 2>&1 | tail -3
