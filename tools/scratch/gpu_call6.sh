#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_configs.py tests/test_gpu_nets.py tests/test_gpu_v10.py "tests/test_gpu_conv.py::test_no_generic_fallback_kernel_in_16bit_modes" -m gpu -q -s > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log )
grep -v "^$" $out/pytest.log | grep -v "amdgpu.ids" | grep "S=\|passed\|failed\|FAILED\|Error\|assert\|prob diff" | cut -c1-600 | tail -40
for p in north-star c4 c5 v10; do
( timeout 900 python bench.py --preset $p --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err; echo "bench exit $?" >> $out/bench_$p.err )
tail -2 $out/bench_$p.err; python -c "
import json;d=json.load(open('$out/bench_$p.json'));print('$p',d['value'],d['ms_per_step'],d['stages'],d['roofline']['all_conv_frac'],d['roofline']['traffic'],d['roofline']['traffic_source']); e=d['parity']['e2e']; print({k:e.get(k) for k in ('frames','frac_identical_candidate_sets','frac_identical_survivor_sets','frac_equivalent_survivor_sets','frac_identical_survivors_in_order','frac_identical_track_ids','candidate_anchors_differing','candidates_compared','survivor_anchors_differing','survivors_compared','lane_points_off_by_more_than_1px','lane_points_compared','error')}); print(d['config']['candidates_per_frame'], d['config']['detections_per_frame'], d['config']['detections_over_0.6'], d['config']['tracked_per_stream'])"
done
