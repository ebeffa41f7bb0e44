#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_v10.py -m gpu -q > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log ); tail -3 $out/pytest.log
ADAS_NO_DW_SLIDING=1 python tools/profile_layers.py yolov10n --batch 64 --precision fp16 --top 120 2>&1 | grep "dwconv\|batch" > $out/dw_old.txt; head -1 $out/dw_old.txt
python tools/profile_layers.py yolov10n --batch 64 --precision fp16 --top 120 2>&1 | grep "dwconv\|batch" > $out/dw_new.txt; head -1 $out/dw_new.txt
paste <(awk '{print $1, $8}' $out/dw_old.txt) <(awk '{print $1}' $out/dw_new.txt) | head -20
for v in 1 0 1 0; do ADAS_NO_DW_SLIDING=$v python bench.py --preset v10 --no-extras --no-cpu-baseline --repeats 3 > $out/b_$v.json 2>/dev/null; python -c "
import json;d=json.load(open('$out/b_$v.json'));print('no_sliding=$v',d['value'],d['repeats']['fps_median'],d['stages']['det_net_ms'])"; done
