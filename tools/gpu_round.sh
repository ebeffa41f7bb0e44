#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_round.sh <tag>
# runs the gpu tests, the default bench, and a rocprofv3 --kernel-trace --stats pass of the same bench command;
# everything lands under gpurun_out/<tag>/
tag=${1:-r02}
commit=${2:-unknown}     # the caller passes `git rev-parse --short HEAD` (the GPU box has no .git)
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log )
tail -5 $out/pytest_gpu.log
# HBM traffic of the dominant kernel: two PMC passes (FETCH_SIZE costs 3 of 4 TCC slots)
for C in FETCH_SIZE WRITE_SIZE; do
  ( timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out/pmc_$C -o p -- python bench.py --no-cpu-baseline --no-extras --no-overlap --steps 5 --warmup 2 > /dev/null 2> $out/pmc_$C.err )
done
python tools/traffic_summary.py $out "conv_h8_kernel<adas::Fp16, 2" "conv_h8_kernel<RELU>" 64 fp16 $commit > $out/traffic.json   # both sync variants
cat $out/traffic.json
[ -s $out/traffic.json ] && grep -q hbm_bytes_per_launch $out/traffic.json && cp $out/traffic.json profiles/traffic.json   # bench.py reads it
( timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err )
cat $out/bench.json
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --no-cpu-baseline --no-extras > $out/bench_prof.json 2> $out/bench_prof.err )
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -25 $f
# drop the bulky per-dispatch trace, keep the stats
find $out/prof -name '*kernel_trace.csv' -size +8M -delete
# the same bench with both nets on one stream: per-kernel durations comparable with bench.py's own hipEvent pass
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_noov -o bench -- python bench.py --no-cpu-baseline --no-extras --no-overlap > $out/bench_noov.json 2> $out/bench_noov.err )
f=$(find $out/prof_noov -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats_no_overlap.csv
find $out/prof_noov -name '*kernel_trace.csv' -size +8M -delete
find $out -name '*kernel_trace.csv' -size +8M -delete
