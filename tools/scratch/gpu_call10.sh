#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03j
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_silu.txt 2>&1; head -1 $out/layers_silu.txt
ADAS_DEBUG_RELU_FOR_SILU=1 python tools/profile_layers.py yolov8n --batch 64 --precision fp16 --top 100 > $out/layers_relu.txt 2>&1; head -1 $out/layers_relu.txt
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r"\s*([\d.]+) ms\s+[\d.]+%\s+([\d.]+) GF\s+([\d.]+) TF/s\s+(\S+)\s+(.*)",l)
        if m: d[m.group(4)]=(float(m.group(1)),m.group(5).strip())
    return d
a=load("gpurun_out/r03j/layers_silu.txt"); b=load("gpurun_out/r03j/layers_relu.txt")
tot=0
for k,(ms,desc) in sorted(a.items(), key=lambda kv:-kv[1][0]):
    if k in b:
        tot+=ms-b[k][0]
        print("%-28s silu %.4f relu %.4f  d %.4f  %s" % (k, ms, b[k][0], ms-b[k][0], desc[:60]))
print("total silu-relu", tot)
PY
