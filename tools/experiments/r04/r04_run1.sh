#!/bin/bash
# round-4 GPU call 1: split-precision (fp16x3) correctness + first timings + the small-map PMC evidence
out=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_x3.py -q -s -x > $out/pytest_x3.log 2>&1; echo "exit $?" >> $out/pytest_x3.log ); tail -3 $out/pytest_x3.log
( timeout 900 python -m pytest tests/test_gpu_chain.py -q -s -k "x3 or split or exactly or exact" > $out/pytest_chain_x3.log 2>&1; echo "exit $?" >> $out/pytest_chain_x3.log ); tail -3 $out/pytest_chain_x3.log
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 40 > $out/layers_ufld_x3.txt 2>&1
python tools/profile_layers.py yolov8n --batch 64 --precision fp16x3 --top 80 > $out/layers_v8n_x3.txt 2>&1
head -3 $out/layers_ufld_x3.txt; head -3 $out/layers_v8n_x3.txt
( timeout 600 python bench.py --precision fp16x3 --no-cpu-baseline --no-extras --steps 10 --repeats 1 > $out/bench_x3.json 2> $out/bench_x3.err; echo "exit $?" >> $out/bench_x3.err )
cut -c1-400 $out/bench_x3.json; tail -2 $out/bench_x3.err
# PMC: conv_halo<64,SILU,s1> and conv_pw<12> on 40x40 maps at 64 frames (VERDICT r3 item 2's missing evidence)
P="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU"
cd /tmp
rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_halo -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --hw 40 40 --cin 64 --cout 64 --k 3 --s 1 --batch 64 --precision fp16 --act 1 --iters 5 > $out/pmc_halo.log 2>&1
rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc_pw -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py --hw 40 40 --cin 384 --cout 128 --k 1 --s 1 --batch 64 --precision fp16 --act 1 --iters 5 > $out/pmc_pw.log 2>&1
cd $GRAFT_REPO_ROOT
( python tools/pmc_summary.py $out/pmc_halo conv_; python tools/pmc_summary.py $out/pmc_pw conv_ ) > $out/pmc_halo_40x40.txt 2>&1
cat $out/pmc_halo_40x40.txt | head -40
find $out -name '*kernel_trace.csv' -delete; find $out -name '*agent_info.csv' -delete; find $out -name '*counter_collection.csv' -size +2M -delete
