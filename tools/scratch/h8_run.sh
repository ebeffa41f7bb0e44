cd $GRAFT_REPO_ROOT
cp vehicle-cv-adas_amd/libadas_hip.so /tmp/lib_keep.so
for v in base noprio look5 base noprio look5; do cp vehicle-cv-adas_amd/_ab/lib_$v.so vehicle-cv-adas_amd/libadas_hip.so; echo "== $v $(timeout 300 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 30 2>/dev/null | grep -E "layer[234]\.[01]\.conv[12] .*k3s1" | awk '{s+=$1} END {print "sum9", s}')"; done
cp /tmp/lib_keep.so vehicle-cv-adas_amd/libadas_hip.so
