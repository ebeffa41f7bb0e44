cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -2
for L in "40 200 128 128 1" "20 100 256 256 1" "10 50 512 512 1" "20 100 256 256 2" "10 50 512 512 2" ; do
  set -- $L
  for g in 1 0 2 8; do
    echo -n "CBG=$g  "; ADAS_HALO_CBG=$g timeout 120 python tools/bench_conv.py --hw $1 $2 --cin $3 --cout $4 --k 3 --s $5 --batch 64 --iters 30 2>&1 | tail -1
  done
done
for L in "80 80 64 64" "40 40 128 128" "20 20 256 256" "80 80 32 64"; do
  set -- $L
  for g in 1 0; do
    echo -n "CBG=$g  "; ADAS_HALO_CBG=$g timeout 120 python tools/bench_conv.py --hw $1 $2 --cin $3 --cout $4 --k 3 --s 1 --batch 64 --iters 30 --act 1 2>&1 | tail -1
  done
done
