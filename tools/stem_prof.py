#!/usr/bin/env python3
"""scratch (round 5): where a tile of the fused ResNet stem (conv_stem_kernel<7,4,RELU,POOL>) spends its cycles -- wave 0's shader clock
per phase, summed over workgroups.  Needs ADAS_LIB=<library whose conv_stem.hip was built with -DADAS_STEM_PROF>:
    python tools/stem_prof.py [--model ufldv2_res18] [--batch 64]"""
import argparse, ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
import netutil
CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="ufldv2_res18"); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--prec", default="fp16")
a = ap.parse_args()
path, W, g = netutil.model(a.model)
e = CE.HipEngine(path, a.prec, a.batch)
shp = e.get_engine_input_shape()
x = np.random.default_rng(0).uniform(0, 1, (a.batch,) + tuple(shp[1:])).astype(np.float32)
buf = L.DeviceBuffer.from_array(x)
lib = C.CDLL(L.LIB_PATH)
prof = hasattr(lib, "adas_debug_stem_prof")
if not prof:
    lib.adas_debug_stem_prof = lambda *a: 0     # plain library: the timings only
out = (C.c_ulonglong * 16)()
rows = e.profile(buf.ptr, a.batch, 5)
# the pipeline's seam: the (c0,c1,c2,0) 16-bit NHWC tensor of adas_preprocess_*_packed
xp = np.zeros((a.batch, shp[2], shp[3], 4), np.float16)
xp[..., :3] = x.transpose(0, 2, 3, 1)
pbuf = L.DeviceBuffer.from_array(xp)
e.infer_device_packed(pbuf.ptr, a.batch); L.lib().adas_synchronize()
lib.adas_debug_stem_prof(out, 1)
import time
best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    for _ in range(5 if prof and rep == 0 else 20):
        e.infer_device_packed(pbuf.ptr, a.batch)
    L.lib().adas_synchronize()
    best = min(best, (time.perf_counter() - t0) / (5 if prof and rep == 0 else 20) * 1e3)
    if prof:
        break
print("%s: packed input %.3f ms per forward (best of 4 x 20); fp32 seam per-layer pass: stem %.3f ms" % (os.path.basename(L.LIB_PATH), best, rows[1][3]))
lib.adas_debug_stem_prof(out, 0)
names = ["(loop overhead)", "barrier: previous tile done", "window regs -> LDS (waits for the loads)", "barrier", "next tile's loads issued", "MFMA K loop",
         "barrier: window reads done", "bias + act + conv tile -> LDS", "barrier", "pool + store"]
print("per-layer pass (fp32 seam):", [(r[0], round(r[3], 4)) for r in rows[:3]])
if not prof:
    sys.exit(0)
tiles = out[15]
tot = sum(out[:10])
print("%s batch %d: %d tiles sampled, %.0f cycles per tile" % (a.model, a.batch, tiles, tot / max(tiles, 1)))
for i, n in enumerate(names):
    print("  %-45s %8.0f cycles  %5.1f %%" % (n, out[i] / max(tiles, 1), 100.0 * out[i] / max(tot, 1)))
e.close()
