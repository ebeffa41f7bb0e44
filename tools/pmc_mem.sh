#!/bin/bash
# usage: tools/pmc_mem.sh <outdir> [bench_conv args...]  -- memory-side PMC passes (separate passes: TCC has 4 slots, FETCH_SIZE costs 3)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/m$i -o m$i -- python tools/bench_conv.py --iters 5 "$@" > $out.mlog$i 2>&1
done
python tools/pmc_summary.py $out conv_
