#!/bin/bash
# round-4 GPU call 2: conv_h8x3 correctness (both synchronisation variants) + timings, layer drift of YOLOv8l in fp16
out=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_x3.py -q -s > $out/pytest_x3.log 2>&1; echo "exit $?" >> $out/pytest_x3.log ); tail -3 $out/pytest_x3.log
( ADAS_HALO8_X3=3 timeout 600 python -m pytest tests/test_gpu_x3.py -q -s -k "halo8 or bench_batch" > $out/pytest_x3_mode1.log 2>&1; echo "exit $?" >> $out/pytest_x3_mode1.log ); tail -3 $out/pytest_x3_mode1.log
python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 40 > $out/layers_ufld_x3.txt 2>&1
ADAS_HALO8_X3=3 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 40 > $out/layers_ufld_x3_mode1.txt 2>&1
ADAS_HALO8_X3=2 python tools/profile_layers.py ufldv2_res18 --batch 64 --precision fp16x3 --top 40 > $out/layers_ufld_x3_mode2.txt 2>&1
python tools/profile_layers.py yolov8n --batch 64 --precision fp16x3 --top 80 > $out/layers_v8n_x3.txt 2>&1
head -20 $out/layers_ufld_x3.txt | cut -c1-140; head -3 $out/layers_ufld_x3_mode1.txt; head -3 $out/layers_ufld_x3_mode2.txt;  head -3 $out/layers_v8n_x3.txt
( timeout 600 python bench.py --precision fp16x3 --no-cpu-baseline --no-extras --steps 10 --repeats 1 > $out/bench_x3.json 2> $out/bench_x3.err; echo "exit $?" >> $out/bench_x3.err )
cut -c1-300 $out/bench_x3.json; tail -2 $out/bench_x3.err
python tools/layer_drift.py yolov8l fp16 > $out/layer_drift_yolov8l.txt 2>&1; tail -5 $out/layer_drift_yolov8l.txt
