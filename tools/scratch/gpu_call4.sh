#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03d
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_chain.py "tests/test_gpu_conv.py::test_wide_pointwise_conv_gemm_kernel" "tests/test_gpu_conv.py::test_wide_pointwise_conv_with_residual" "tests/test_gpu_conv.py::test_no_generic_fallback_kernel_in_16bit_modes" tests/test_gpu_configs.py tests/test_gpu_frontend.py tests/test_gpu_pipeline.py -m gpu -q -s > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log )
grep -v "^$" $out/pytest.log | grep -v "amdgpu.ids" | grep "S=\|passed\|failed\|FAILED\|Error\|assert" | cut -c1-900 | tail -40
for p in c4 c5; do
( timeout 900 python bench.py --preset $p --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err; echo "bench exit $?" >> $out/bench_$p.err )
tail -2 $out/bench_$p.err; python -c "
import json;d=json.load(open('$out/bench_$p.json'));print(d['value'],d['ms_per_step'],d['stages'],d['roofline']['all_conv_frac']); print([(k['kernel'],k['ms'],k['tflops']) for k in d['roofline']['top_kernels']]); e=d['parity']['e2e']; print({k:e.get(k) for k in ('frames','frac_identical_candidate_sets','frac_identical_survivor_sets','frac_identical_track_ids','candidate_anchors_differing','candidates_compared','survivor_anchors_differing','survivors_compared')})"
done
( timeout 600 python bench.py --preset c5 --micro-batch 48 --no-cpu-baseline --no-extras > $out/bench_c5_b48.json 2> $out/bench_c5_b48.err )
python -c "
import json;d=json.load(open('$out/bench_c5_b48.json'));print('c5 B=48',d['value'],d['ms_per_step'],d['roofline']['all_conv_frac']); print([(k['kernel'],k['ms'],k['tflops']) for k in d['roofline']['top_kernels']])"
