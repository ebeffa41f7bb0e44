#!/usr/bin/env python3
"""Device time of the two pre-processing kernels (packed fp16 seam) at S frames of 1280x720: python tools/bench_pre.py [S]
ADAS_PRE_ROWS=<rows per workgroup> is read by the library at first launch (one process per value)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
L = importlib.import_module("adas_amd._lib")
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cam = bench.cam_frames(S, 3)
dc = L.DeviceBuffer.from_array(cam)
ty = L.DeviceBuffer(S * 640 * 640 * 8); tl = L.DeviceBuffer(S * 320 * 1600 * 8)
lib = L.lib()
def run_y(): L.check(lib.adas_preprocess_yolo_packed_prec(dc.ptr, S, 720, 1280, ty.ptr, 640, 640, 1, L.PREC_FP16, None))
def run_l(): L.check(lib.adas_preprocess_ufld_packed_prec(dc.ptr, S, 720, 1280, tl.ptr, 320, 1600, 0.6, L.PREC_FP16, None))
res = {}
for name, fn in (("yolo", run_y), ("ufld", run_l)):
    for _ in range(3): fn()
    with L.StreamTimer(None) as t:
        for _ in range(20): fn()
    res[name] = t.ms / 20 * 1e3
    t.close()
a = ty.download((S, 640, 640, 4), np.uint16); b = tl.download((S, 320, 1600, 4), np.uint16)
print("ADAS_PRE_ROWS=%s  yolo %.1f us  ufld %.1f us  checksums %d %d" % (os.environ.get("ADAS_PRE_ROWS", "8"), res["yolo"], res["ufld"], int(a.astype(np.uint64).sum()), int(b.astype(np.uint64).sum())))
