cd $GRAFT_REPO_ROOT
for u in 3 4 5 24 26 3; do echo "U=$u: $(ADAS_FC_U=$u timeout 300 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 30 2>/dev/null | grep -E "cls.3" | cut -c1-60)"; done
