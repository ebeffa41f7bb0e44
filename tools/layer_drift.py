#!/usr/bin/env python3
"""Per-layer drift of a 16-bit engine precision against the fp32 engine on the same model and frames (GPU):
    python tools/layer_drift.py yolov8n fp16 [bf16 ...]
prints, for every materialised conv activation, rel-L2 and max|diff| against the fp32-mode activation of the same layer."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import netutil
from conftest import load_pkg
load_pkg()
CE = importlib.import_module("adas_amd.coreEngine")
name = sys.argv[1]
precs = sys.argv[2:] or ["fp16"]
path, W, g = netutil.model(name)
x = netutil.lane_frames(2, g.in_h, g.in_w) if name.startswith("ufld") else netutil.coco_like_frames(2, g.in_h, g.in_w)
ref = CE.HipEngine(path, "fp32", 2)
ref.engine_inference(x)
n = ref.stats()["num_layers"]
engs = {p: CE.HipEngine(path, p, 2) for p in precs}
for e in engs.values():
    e.engine_inference(x)
for i in range(n):
    nm, fl, kind = ref.layer_info(i)
    if kind != 1:
        continue
    try:
        a = ref.fetch_activation(i, 2)
    except Exception:
        continue
    row = "%-28s max|ref| %9.3f" % (nm, np.abs(a).max())
    for p, e in engs.items():
        try:
            b = e.fetch_activation(i, 2)
            row += "  | %s %-34s rel %.2e max %.2e" % (p, e.layer_kernel(i, 2)[:34], np.linalg.norm(b - a) / (np.linalg.norm(a) + 1e-30), np.abs(b - a).max())
        except Exception as ex:
            row += "  | %s (fused)" % p
    print(row)
