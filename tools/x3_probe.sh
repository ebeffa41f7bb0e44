#!/bin/bash
# usage (GPU box, via gpurun): tools/x3_probe.sh <tag>  -- where conv_h8x3_kernel spends its time: same-box A/B of its switches on the whole
# bench step, per-phase cycle counters (scratch build -DADAS_H8X_PROF) and PMC passes on single layers.  Output: gpurun_out/<tag>/
tag=${1:-x3probe}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# 1. A/B of the kernel's switches on the north-star step (exact mode): LH = L half-chunks fetch their cross rows only;
#    ADAS_HALO8_X3 = 2 / 3 forces the one-barrier-per-tap-row / wave-groups-a-barrier-apart variant on every layer
( timeout 1200 python tools/ab_bench.py --rounds 2 --steps 30 --repeats 2 --variant base --variant lh0:ADAS_H8X_LHALF=0 --variant mode3:ADAS_HALO8_X3=3 \
    --variant mode2:ADAS_HALO8_X3=2 --variant lh0mode2:ADAS_H8X_LHALF=0,ADAS_HALO8_X3=2 > $out/ab_h8x3.txt 2>&1 ); tail -8 $out/ab_h8x3.txt
# 2. per-layer tables of both nets at HEAD (exact mode)
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 40 > $out/layers_ufldv2_res18_b64_fp16x3.txt 2>&1
python tools/profile_layers.py yolov8n --batch 64 --precision fp16x3 --top 100 > $out/layers_yolov8n_b64_fp16x3.txt 2>&1
head -3 $out/layers_ufldv2_res18_b64_fp16x3.txt | cut -c1-140
# 3. phase counters (instrumented scratch library)
P=vehicle-cv-adas_amd/_scratch/libadas_hip_h8xprof.so
if [ -f $P ]; then
  for args in "--hw 80 400 --cin 64 --cout 64" "--hw 80 400 --cin 64 --cout 64 --res" "--hw 40 200 --cin 128 --cout 128 --res" "--hw 20 100 --cin 256 --cout 256" "--hw 10 50 --cin 512 --cout 512"; do
    ADAS_LIB=$P timeout 200 python tools/experiments/h8x_prof.py $args --batch 64 >> $out/h8x_phases.txt 2>&1
  done
  cat $out/h8x_phases.txt | head -40
fi
# 4. PMC on two single layers
for shape in "80 400 64 64" "40 200 128 128"; do
  set -- $shape
  timeout 500 tools/pmc_conv.sh $out/pmc_$3 --hw $1 $2 --cin $3 --cout $4 --batch 64 --precision fp16x3 > $out/pmc_h8x3_$3.txt 2>&1
  rm -rf $out/pmc_$3
done
grep -c . $out/pmc_h8x3_64.txt
