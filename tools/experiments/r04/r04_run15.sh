#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04o
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log ); tail -6 $out/pytest_gpu.log
# layer1 with the tensors cache-resident (batch 8: 33 MB per tensor, inside the 256 MB MALL) vs streamed from HBM (batch 64)
for b in 8 16 64; do
python tools/profile_layers.py ufldv2_res18 --batch $b --precision fp16 --top 100 > $out/layers_ufldv2_res18_b${b}_fp16.txt 2>&1; head -1 $out/layers_ufldv2_res18_b${b}_fp16.txt; grep "layer1\|conv1 " $out/layers_ufldv2_res18_b${b}_fp16.txt | cut -c1-150
done
