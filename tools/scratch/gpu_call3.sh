#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_v10.py tests/test_gpu_chain.py -m gpu -q -s > $out/pytest_v10.log 2>&1; echo "exit $?" >> $out/pytest_v10.log )
grep -v "^$" $out/pytest_v10.log | grep -v "amdgpu.ids" | cut -c1-700 | tail -50
( timeout 600 python bench.py --preset v10 --no-cpu-baseline > $out/bench_v10.json 2> $out/bench_v10.err; echo "bench exit $?" >> $out/bench_v10.err )
tail -3 $out/bench_v10.err; python -c "
import json;d=json.load(open('$out/bench_v10.json'));print(d['value'],d['stages'],d['parity']['e2e']); print([(k['kernel'],k['ms'],k['tflops']) for k in d['roofline']['top_kernels']])"
python tools/profile_layers.py yolov10n --batch 64 --precision fp16 --top 100 > $out/layers_yolov10n_b64_fp16.txt 2>&1; head -40 $out/layers_yolov10n_b64_fp16.txt
