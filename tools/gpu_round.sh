#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_round.sh <tag>
# runs the gpu tests, the default bench, and a rocprofv3 --kernel-trace --stats pass of the same bench command;
# everything lands under gpurun_out/<tag>/
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log )
tail -5 $out/pytest_gpu.log
( timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err )
cat $out/bench.json
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err )
f=$(find $out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -25 $f
# drop the bulky per-dispatch trace, keep the stats
find $out/prof -name '*kernel_trace.csv' -size +8M -delete
