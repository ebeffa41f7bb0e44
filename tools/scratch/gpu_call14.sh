#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03n
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_v10.py tests/test_gpu_configs.py -m gpu -q -s > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log ); grep -v "^$" $out/pytest.log | grep "model.2 out\|passed\|failed\|FAILED\|Error" | cut -c1-300 | tail -20
ADAS_NO_C2F_FUSE=1 python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 12 2>&1 | grep "model.2\|batch" 
python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 2>&1 | grep "model.2\.\|batch"
for v in 1 0 1 0; do ADAS_NO_C2F_FUSE=$v python bench.py --no-extras --no-cpu-baseline --repeats 3 > $out/b_$v.json 2>/dev/null; python -c "
import json;d=json.load(open('$out/b_$v.json'));print('no_c2f_fuse=$v',d['value'],d['repeats']['fps_median'],d['stages']['det_net_ms'],d['config']['kernel_launches_per_step_nets'])"; done
