cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -m gpu -x -q 2>&1 | tail -2
for c in 0 1; do
  echo "== RW_CONTIG=$c STEM_CONTIG=$c"
  ADAS_RW_CONTIG=$c ADAS_STEM_CONTIG=$c timeout 200 python tools/profile_layers.py ufldv2_res18 --batch 64 --top 8 2>&1 | head -12
  ADAS_RW_CONTIG=$c ADAS_STEM_CONTIG=$c timeout 200 python tools/profile_layers.py yolov8n --batch 64 --top 6 2>&1 | head -9
done
