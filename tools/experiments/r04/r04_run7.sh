#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_post.py tests/test_gpu_frontend.py tests/test_gpu_chain.py -q -k "spill or bytetracker or fp16 or overflow" > $out/pytest_new.log 2>&1; echo "exit $?" >> $out/pytest_new.log ); tail -4 $out/pytest_new.log
python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_yolov8n_b64_fp16.txt 2>&1; head -3 $out/layers_yolov8n_b64_fp16.txt
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16 --top 100 > $out/layers_ufldv2_res18_b64_fp16.txt 2>&1; head -2 $out/layers_ufldv2_res18_b64_fp16.txt
P="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU"
cd /tmp
for prec in fp16 fp16x3; do
  ADAS_BENCH_NO_PMC=1 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_$prec -o p -- python $GRAFT_REPO_ROOT/bench.py --precision $prec --no-cpu-baseline --no-extras --no-overlap --steps 3 --warmup 1 --repeats 0 --latency-steps 8 > $out/pmc_$prec.json 2> $out/pmc_$prec.err
  python $GRAFT_REPO_ROOT/tools/pmc_top.py $out/pmc_$prec 8 > $out/pmc_top_kernels_$prec.txt 2>&1
done
cd $GRAFT_REPO_ROOT
head -24 $out/pmc_top_kernels_fp16.txt
find $out -name '*kernel_trace.csv' -delete; find $out -name '*agent_info.csv' -delete; find $out -name '*counter_collection.csv' -delete
