cd $GRAFT_REPO_ROOT
B=tools/experiments/_bin
for cfg in "16 60 120 256 512" "64 20 120 256 512" "16 150 100 256 512" "4 400 60 512 512" "2 600 40 1024 1024"; do
  for v in bt_old bt_new bt_new2; do timeout 120 $B/$v $cfg; done
done
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_post.py tests/test_gpu_frontend.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ud -o b -- python bench.py --no-cpu-baseline --no-overlap --steps 20 > gpurun_out/ud_bench.json 2>/dev/null; f=$(find gpurun_out/prof_ud -name "*kernel_stats.csv" | head -1); grep -i "ufld_decode\|yolo_post\|bytetrack\|detect_v8\|yolo_scan" $f | cut -d, -f1-4; find gpurun_out/prof_ud -name "*kernel_trace.csv" -delete
timeout 200 python bench.py --det yolov8s --streams 16 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stages'], d['config']['detections_per_frame'])"
