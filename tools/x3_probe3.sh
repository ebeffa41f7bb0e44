#!/bin/bash
# usage (GPU box): tools/x3_probe3.sh <tag> -- correctness + same-box A/B of compile-time variants of conv_h8x3 (scratch libraries under _scratch/)
tag=${1:-x3probe3}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/vehicle-cv-adas_amd/_scratch
( timeout 600 python -m pytest tests/test_gpu_x3.py -m gpu -q -x > $out/pytest_x3.log 2>&1 ); tail -2 $out/pytest_x3.log
V="--variant base"
for t in "$@"; do [ -f $S/libadas_hip_$t.so ] && V="$V --variant $t:ADAS_LIB=$S/libadas_hip_$t.so"; done
( timeout 1200 python tools/ab_bench.py --rounds 3 --steps 30 --repeats 2 $V > $out/ab_variants.txt 2>&1 ); tail -8 $out/ab_variants.txt
