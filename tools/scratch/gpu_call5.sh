#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03e
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_v10.py "tests/test_gpu_conv.py::test_no_generic_fallback_kernel_in_16bit_modes" tests/test_gpu_nets.py -m gpu -q -s > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log )
grep -v "^$" $out/pytest.log | grep -v "amdgpu.ids" | grep "S=\|passed\|failed\|FAILED\|Error\|assert\|yolov10n fp16 head" | cut -c1-700 | tail -30
for p in north-star c4 v10; do
( timeout 900 python bench.py --preset $p --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err; echo "bench exit $?" >> $out/bench_$p.err )
tail -2 $out/bench_$p.err; python -c "
import json;d=json.load(open('$out/bench_$p.json'));print('$p',d['value'],d['ms_per_step'],d['stages'],d['roofline']['all_conv_frac']); e=d['parity']['e2e']; print({k:e.get(k) for k in ('frames','frac_identical_candidate_sets','frac_identical_survivor_sets','frac_identical_survivors_in_order','frac_identical_track_ids','candidate_anchors_differing','candidates_compared','survivor_anchors_differing','survivors_compared','lane_points_off_by_more_than_1px','lane_points_compared')}); print(d['config']['candidates_per_frame'], d['config']['detections_per_frame'], d['unfiltered'])"
done
