cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv.py -q -k "c2f" 2>&1 | tail -12
timeout 300 python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 70 2>/dev/null | grep -E "ms/step|model.2\.|model.4\.|model.15\."
