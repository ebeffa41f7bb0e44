cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_configs.py -q 2>&1 | tail -4
timeout 300 python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 80 2>/dev/null | grep -E "ms/step|model.1[0235]\.cv1|model.1[03] "
