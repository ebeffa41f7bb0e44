cd $GRAFT_REPO_ROOT
for nh in 0 1; do echo "== NO_HALO=$nh"; ADAS_NO_HALO=$nh timeout 300 python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 80 2>/dev/null | grep -E "ms/step|model\.(6|8|12|18|21)\.m\.0\.cv1|model.22.cv2.2|model.22.cv2.1.1"; done
