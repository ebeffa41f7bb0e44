#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03k
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -m gpu -q -x > $out/pytest.log 2>&1; echo "exit $?" >> $out/pytest.log ); tail -3 $out/pytest.log
ADAS_HALO_BM128=0 python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_bm256.txt 2>&1; head -1 $out/layers_bm256.txt
python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_bm128.txt 2>&1; head -1 $out/layers_bm128.txt
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*([\d.]+) ms\s+[\d.]+%\s+([\d.]+) GF\s+([\d.]+) TF/s\s+(\S+)\s+(.*)",l)
        if m: d[m.group(4)]=(float(m.group(1)),m.group(5).strip())
    return d
a=load("gpurun_out/r03k/layers_bm256.txt"); b=load("gpurun_out/r03k/layers_bm128.txt")
tot=0
for k,(ms,desc) in sorted(b.items(), key=lambda kv:-kv[1][0]):
    if "bm128" in desc:
        tot+=a[k][0]-ms
        print("%-28s bm256 %.4f bm128 %.4f  gain %.4f  %s" % (k, a[k][0], ms, a[k][0]-ms, desc[:64]))
print("total gain", tot)
PY
for v in 0 1 0 1; do ADAS_HALO_BM128=$v python bench.py --no-extras --no-cpu-baseline --repeats 3 > $out/b_$v.json 2>/dev/null; python -c "
import json;d=json.load(open('$out/b_$v.json'));print('bm128=$v',d['value'],d['repeats']['fps_median'],d['stages']['det_net_ms'])"; done
ADAS_HALO_BM128=0 python bench.py --preset c5 --micro-batch 1 --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('c5 B=1 bm256',d['value'],d['stages'])"
python bench.py --preset c5 --micro-batch 1 --no-extras --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('c5 B=1 bm128',d['value'],d['stages'])"
