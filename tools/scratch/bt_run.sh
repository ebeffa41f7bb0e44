cd $GRAFT_REPO_ROOT
B=tools/scratch/_bin
for cfg in "16 60 120 256 512" "64 20 120 256 512" "16 150 100 256 512" "4 400 60 512 512" "2 600 40 1024 1024"; do
  for v in bt_old bt_new; do timeout 120 $B/$v $cfg; done
done
timeout 120 $B/bt_old_prof 16 60 120 256 512
timeout 120 $B/bt_new_prof 16 60 120 256 512
timeout 120 $B/bt_new_prof 16 150 100 256 512
timeout 600 python -m pytest tests/test_gpu_post.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --det yolov8s --streams 16 --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('stages') or d['config'])"
