#!/bin/bash
# usage (GPU box, via gpurun): tools/r04_round.sh <commit>  -- the round-4 evidence set under gpurun_out/r04/
commit=${1:-unknown}
out=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "commit $commit" > $out/commit.txt
( timeout 1200 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log ); tail -3 $out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke exit $?" >> $out/smoke.log ); tail -2 $out/smoke.log
( timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err ); cut -c1-400 $out/bench.json
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --no-cpu-baseline --no-extras > $out/bench_prof.json 2> $out/bench_prof.err )
f=$(find $out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && head -8 $f | cut -c1-160
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_noov -o bench -- python bench.py --no-cpu-baseline --no-extras --no-overlap > $out/bench_noov.json 2> $out/bench_noov.err )
f=$(find $out/prof_noov -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_no_overlap.csv
if [ "$2" = "x3prof" ]; then
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_x3 -o bench -- python bench.py --precision fp16x3 --no-cpu-baseline --no-extras --no-overlap > $out/bench_x3_noov.json 2> $out/bench_x3.err )
f=$(find $out/prof_x3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_fp16x3_no_overlap.csv
fi
for p in c4 v9 v7 v6; do
  ( ADAS_BENCH_NO_PMC=1 timeout 300 python bench.py --preset $p --no-cpu-baseline --steps 20 --repeats 2 > $out/bench_$p.json 2> /dev/null )
  python -c "
import json; d=json.load(open('$out/bench_$p.json')); print('$p', d['value'], d['ms_per_step'])"
done
rm -rf $out/prof $out/prof_noov $out/prof_x3
( timeout 300 python bench.py --preset c5 --micro-batch 1 --no-cpu-baseline --no-extras --steps 200 --repeats 2 > $out/bench_c5_frame_at_a_time.json 2>/dev/null )
( timeout 400 python bench.py --preset c5 --no-cpu-baseline > $out/bench_c5.json 2> /dev/null )
for p in c5_frame_at_a_time c5; do python -c "
import json; d=json.load(open('$out/bench_$p.json')); print('$p', d['value'], d['ms_per_step'])"; done
python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_yolov8n_b64_fp16.txt 2>&1
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 100 > $out/layers_ufldv2_res18_b64_fp16.txt 2>&1
python tools/profile_layers.py yolov8l --batch 1 --precision fp16 --top 200 > $out/layers_yolov8l_b1_fp16.txt 2>&1
python tools/profile_layers.py efficientdet-d0 --batch 64 --precision fp16 --top 300 > $out/layers_efficientdet-d0_b64_fp16.txt 2>&1; head -6 $out/layers_efficientdet-d0_b64_fp16.txt | cut -c1-150
P="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU"
cd /tmp
for prec in fp16; do
  ADAS_BENCH_NO_PMC=1 timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$prec -o p -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --no-cpu-baseline --no-extras --no-overlap --steps 3 --warmup 1 --repeats 0 --latency-steps 8 > $out/pmc_$prec.json 2> $out/pmc_$prec.err
  python $GRAFT_REPO_ROOT/tools/pmc_top.py $out/pmc_$prec 16 > $out/pmc_top_kernels_$prec.txt 2>&1
  rm -rf $out/pmc_$prec
done
head -16 $out/pmc_top_kernels_fp16.txt | cut -c1-200
