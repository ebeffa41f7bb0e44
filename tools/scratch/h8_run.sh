cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -k "projection_shortcut or dma_fed" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_configs.py -q -k "bench_batch or persistent" 2>&1 | tail -1
for m in 0 0; do timeout 300 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 30 2>/dev/null | grep -E "ms/step|layer[234]\.0\.conv2"; done
