#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03q
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_v9.py "tests/test_gpu_conv.py::test_no_generic_fallback_kernel_in_16bit_modes" -m gpu -q -s > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log ); grep -v "^$" $out/pytest.log | grep "yolov9t\|passed\|failed\|FAILED\|Error\|assert" | cut -c1-260 | tail -30
( timeout 600 python bench.py --preset v9 --no-cpu-baseline > $out/bench_v9.json 2> $out/bench_v9.err; echo "bench exit $?" >> $out/bench_v9.err ); tail -2 $out/bench_v9.err
python -c "
import json;d=json.load(open('$out/bench_v9.json'));print('v9',d['value'],d['ms_per_step'],d['stages'],d['roofline']['all_conv_frac'],d['config']['kernel_launches_per_step_nets']); e=d['parity']['e2e']; print({k:e.get(k) for k in ('frames','frac_identical_candidate_sets','frac_identical_survivor_sets','frac_identical_track_ids','error')}); print({k:d['parity'][k] for k in d['parity'] if k.startswith('det_')})"
python tools/profile_layers.py yolov9t --batch 64 --precision fp16 --top 25 > $out/layers_yolov9t_b64_fp16.txt 2>&1; head -26 $out/layers_yolov9t_b64_fp16.txt
