cd $GRAFT_REPO_ROOT
for sn in 0 1 0 1; do echo "== snake $sn"; ADAS_SNAKE=$sn timeout 300 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 30 2>/dev/null | grep -E "ms/step|layer1|layer2\.[01]\.conv[12] .*k3s1"; done
