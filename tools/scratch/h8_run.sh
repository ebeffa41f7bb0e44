cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "dma_fed" 2>&1 | tail -1
for m in 1 1; do echo "== mode $m"; ADAS_HALO8=$m timeout 300 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 20 2>/dev/null | grep -E "ms/step|layer[234]\.[01]\.conv[12] .*k3s1"; done
