cd $GRAFT_REPO_ROOT
for L in "40 200 128 128 1 2" "20 100 256 256 1 2" "10 50 512 512 1 2" "20 100 256 256 2 2" "80 80 64 64 1 1" "40 40 128 128 1 1" "160 160 32 32 1 1"; do
  set -- $L
  for x in 0 1; do
    echo -n "XMAP=$x  "; ADAS_HALO_XMAP=$x timeout 120 python tools/bench_conv.py --hw $1 $2 --cin $3 --cout $4 --k 3 --s $5 --batch 64 --iters 30 --act $6 2>&1 | tail -1
  done
done
