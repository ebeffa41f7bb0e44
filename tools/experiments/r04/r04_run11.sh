#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04k
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sw in 0 32 48 64; do
  echo "== ADAS_H8_SW=$sw"
  ADAS_H8_SW=$sw python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 40 2>&1 | grep -E "batch 64|conv_h8" | cut -c1-130 | tee $out/h8_sw$sw.txt
done
