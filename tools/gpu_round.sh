#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_round.sh <tag> <commit>
# the whole GPU suite, the default bench (its roofline.traffic comes from its own rocprofv3 --pmc child runs), rocprofv3
# --kernel-trace --stats passes of the same bench command (overlapped and one-stream), the preset benches, per-layer tables;
# everything lands under gpurun_out/<tag>/
tag=${1:-r03}
commit=${2:-unknown}     # the caller passes `git rev-parse --short HEAD` (the GPU box has no .git)
lite=${3:-full}          # "lite": skip the c4 / c5 / v10 preset benches and the drift tables (GPU-minute budget)
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "commit $commit" > $out/commit.txt
( timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log )
tail -5 $out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke exit $?" >> $out/smoke.log ); tail -2 $out/smoke.log
( timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err )
cat $out/bench.json | cut -c1-600
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --no-cpu-baseline --no-extras > $out/bench_prof.json 2> $out/bench_prof.err )
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -12 $f
# the same bench with both nets on one stream: per-kernel durations comparable with bench.py's own hipEvent pass
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_noov -o bench -- python bench.py --no-cpu-baseline --no-extras --no-overlap > $out/bench_noov.json 2> $out/bench_noov.err )
f=$(find $out/prof_noov -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats_no_overlap.csv
find $out -name '*kernel_trace.csv' -delete
find $out -name '*agent_info.csv' -delete
if [ "$lite" != "lite" ]; then
( timeout 600 python bench.py --preset c5 --micro-batch 48 --no-cpu-baseline --no-extras > $out/bench_c5_b48.json 2> /dev/null )
( timeout 600 python bench.py --preset c5 --micro-batch 1 --no-cpu-baseline --no-extras > $out/bench_c5_b1.json 2> /dev/null )
for p in c4 c5 v10; do
  ( timeout 900 python bench.py --preset $p --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err; echo "bench exit $?" >> $out/bench_$p.err )
  tail -1 $out/bench_$p.err
done
python tools/profile_layers.py yolov10n --batch 64 --precision fp16 --top 120 > $out/layers_yolov10n_b64_fp16.txt 2>&1
python tools/layer_drift.py yolov8n fp16 > $out/layer_drift_yolov8n.txt 2>&1
python tools/layer_drift.py yolov8s fp16 > $out/layer_drift_yolov8s.txt 2>&1
tail -3 $out/layer_drift_yolov8n.txt
fi
python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_yolov8n_b64_fp16.txt 2>&1
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 100 > $out/layers_ufldv2_res18_b64_fp16.txt 2>&1
python tools/profile_layers.py yolov7-tiny --batch 64 --precision fp16 --top 100 > $out/layers_yolov7-tiny_b64_fp16.txt 2>&1
( ADAS_BENCH_NO_PMC=1 timeout 300 python bench.py --preset v7 --no-cpu-baseline > $out/bench_v7.json 2> $out/bench_v7.err; echo "bench exit $?" >> $out/bench_v7.err )
tail -1 $out/bench_v7.err
