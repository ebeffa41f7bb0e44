#!/bin/bash
# usage (GPU box, via gpurun): tools/round.sh <tag> <commit> [stages]   -- the evidence set of a round under gpurun_out/<tag>/
# stages (comma separated, default all): tests,smoke,bench,stats,presets,layers,pmc
tag=${1:-r05}
commit=${2:-unknown}      # the caller passes `git rev-parse --short HEAD` (the GPU box has no .git)
stages=${3:-tests,smoke,bench,stats,presets,layers,pmc}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "commit $commit" > $out/commit.txt
has() { case ",$stages," in *",$1,"*) return 0;; *) return 1;; esac; }
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['dtype'], d['config'].get('stages'), d['config'].get('kernel_launches_per_step_nets'),
          d['roofline'].get('frac'), d['roofline'].get('avg_launch_us_rocprof'), d['roofline'].get('traffic'), d['config'].get('frame_at_a_time'))
except Exception as ex:
    print(sys.argv[1], 'unreadable:', ex)
PY
}
if has tests; then ( timeout 1300 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $out/pytest_gpu.log ); tail -3 $out/pytest_gpu.log; fi
if has smoke; then ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke exit $?" >> $out/smoke.log ); tail -2 $out/smoke.log; fi
if has bench; then ( timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/bench.err ); show $out/bench.json; fi
if has stats; then
  ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --no-cpu-baseline --no-extras > $out/bench_prof.json 2> $out/bench_prof.err )
  f=$(find $out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && head -8 $f | cut -c1-160
  ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_noov -o bench -- python bench.py --no-cpu-baseline --no-extras --no-overlap > $out/bench_noov.json 2> $out/bench_noov.err )
  f=$(find $out/prof_noov -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_no_overlap.csv
  # (the default precision is the exact mode, fp16x3, since round 6: the two runs above are its summaries; this one is the throughput mode's)
  ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_f16 -o bench -- python bench.py --precision fp16 --no-cpu-baseline --no-extras --no-overlap > $out/bench_fp16_noov.json 2> $out/bench_fp16.err )
  f=$(find $out/prof_f16 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_fp16_no_overlap.csv
  rm -rf $out/prof $out/prof_noov $out/prof_f16
fi
if has presets; then
  for p in c4 c5; do ( ADAS_BENCH_NO_PMC=1 timeout 500 python bench.py --preset $p --no-cpu-baseline --steps 20 --repeats 2 > $out/bench_$p.json 2> /dev/null ); show $out/bench_$p.json; done
  ( timeout 500 python bench.py --precision fp16 --no-cpu-baseline --steps 40 --repeats 2 > $out/bench_fp16.json 2> /dev/null ); show $out/bench_fp16.json
  ( ADAS_BENCH_NO_PMC=1 timeout 300 python bench.py --preset c5 --micro-batch 1 --no-cpu-baseline --no-extras --steps 200 --repeats 2 > $out/bench_c5_frame_at_a_time.json 2>/dev/null ); show $out/bench_c5_frame_at_a_time.json
  for p in v10 v9 v7 v6; do ( ADAS_BENCH_NO_PMC=1 timeout 300 python bench.py --preset $p --no-cpu-baseline --no-extras --steps 20 --repeats 2 > $out/bench_$p.json 2> /dev/null ); show $out/bench_$p.json; done
fi
if has layers; then
  for prec in fp16x3 fp16; do
    python tools/profile_layers.py yolov8n --batch 64 --precision $prec --top 100 > $out/layers_yolov8n_b64_$prec.txt 2>&1
    python tools/profile_layers.py ufldv2_res18 --batch 64 --precision $prec --top 100 > $out/layers_ufldv2_res18_b64_$prec.txt 2>&1
  done
  head -4 $out/layers_yolov8n_b64_fp16x3.txt | cut -c1-150
fi
if has pmc; then
  P="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU"
  cd /tmp
  for prec in fp16x3 fp16; do
    ADAS_BENCH_NO_PMC=1 timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$prec -o p -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --no-cpu-baseline --no-extras --no-overlap --steps 3 --warmup 1 --repeats 0 --latency-steps 8 > $out/pmc_$prec.json 2> $out/pmc_$prec.err
    python $GRAFT_REPO_ROOT/tools/pmc_top.py $out/pmc_$prec 24 > $out/pmc_top_kernels_$prec.txt 2>&1
    rm -rf $out/pmc_$prec
  done
  head -12 $out/pmc_top_kernels_fp16x3.txt | cut -c1-200
fi
