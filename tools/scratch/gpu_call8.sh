#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03h
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest "tests/test_gpu_conv.py::test_wide_pointwise_conv_gemm_kernel" "tests/test_gpu_conv.py::test_fp16_stores_saturate_instead_of_overflowing" tests/test_gpu_frontend.py -m gpu -q > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log ); tail -3 $out/pytest.log
for r in 8 16 32 64 160 320; do ADAS_PRE_ROWS=$r python tools/bench_pre.py 64 2>&1 | tail -1 | tee -a $out/pre_rows.txt; done
for r in 8 32 64 8 32; do
  ADAS_PRE_ROWS=$r python bench.py --no-extras --no-cpu-baseline --repeats 3 > $out/b_$r.json 2>/dev/null
  python -c "
import json;d=json.load(open('$out/b_$r.json'));print('rows $r',d['value'],d['repeats'])" | tee -a $out/pre_rows.txt
done
