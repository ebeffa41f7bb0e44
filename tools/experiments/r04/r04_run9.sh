#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for s in 32 24; do
  ADAS_PERSIST_SLOTS_H8=$s rocprofv3 --kernel-trace --output-format csv -d $out/trace$s -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --repeats 0 --latency-steps 8 > $out/t$s.json 2> $out/t$s.err
  echo "== h8 slots $s"; python $GRAFT_REPO_ROOT/tools/trace_overlap.py $out/trace$s 1500 | tee $out/overlap_slots$s.txt
done
find $out -name '*kernel_trace.csv' -delete; find $out -name '*agent_info.csv' -delete
