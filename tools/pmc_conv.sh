#!/bin/bash
# usage: tools/pmc_conv.sh <outdir> [bench_conv args...]   -- PMC passes on the single-layer microbench
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
P4="GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCC_HIT TCC_MISS"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/p$i -o p$i -- python tools/bench_conv.py --iters 5 "$@" > $out.log$i 2>&1
done
python tools/pmc_summary.py $out conv_
