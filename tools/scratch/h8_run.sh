cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -k "stem" 2>&1 | tail -2
timeout 300 python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 80 2>/dev/null | grep -E "ms/step|model.0.conv"
