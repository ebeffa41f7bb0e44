#!/usr/bin/env python3
"""scratch (round 6): per-phase cycle breakdown of conv_h8x3_kernel (the split precision's 3x3 stride-1 conv).  Needs
ADAS_LIB=<a library built with ADAS_BUILD_TAG=h8xprof ADAS_CFLAGS=-DADAS_H8X_PROF python vehicle-cv-adas_amd/build.py>:
   ADAS_LIB=vehicle-cv-adas_amd/_scratch/libadas_hip_h8xprof.so python tools/experiments/h8x_prof.py --hw 80 400 --cin 64 --cout 64 --batch 64 [--res]"""
import argparse, ctypes as C, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
M = importlib.import_module("adas_amd.models"); CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, nargs=2, default=[80, 400]); ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=64); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--res", action="store_true", help="residual add before the activation (a ResNet conv2)")
ap.add_argument("--act", default="relu")
a = ap.parse_args()
H, W = a.hw
ws = M.SynthWeights(0, gain=1.0)
g = M.Graph("unit", 3, H, W, ws)
x, c3 = g.input()
e1 = g.conv(x, a.cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
act = M.ACT_RELU if a.act == "relu" else M.ACT_SILU
use_res = a.res and a.cin == a.cout
y = g.conv(e1, a.cout, 3, 1, "test", act=act, res=(e1 if use_res else None), res_mode=(M.RES_BEFORE_ACT if use_res else M.RES_NONE))
z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
g.output(z, 0, [1, z.h * z.w * 8], "o")
path = os.path.join(tempfile.gettempdir(), "h8x_prof.hipm"); g.save(path)
e = CE.HipEngine(path, "fp16x3", a.batch)
xin = np.random.default_rng(0).uniform(0, 1, (a.batch, 3, H, W)).astype(np.float32)
buf = L.DeviceBuffer.from_array(xin)
lib = C.CDLL(L.LIB_PATH)
prof = hasattr(lib, "adas_debug_h8x_prof")
out = (C.c_ulonglong * 32)()
e.profile(buf.ptr, a.batch, 2)
if prof: lib.adas_debug_h8x_prof(out, 1)
rows = e.profile(buf.ptr, a.batch, 5)
if prof: lib.adas_debug_h8x_prof(out, 0)
li = [i for i, r in enumerate(rows) if r[0] == "test"][0]
ms = rows[li][3]
fl = 2.0 * a.batch * H * W * a.cout * 9 * a.cin
print(f"{H}x{W}x{a.cin}->{a.cout} batch {a.batch}{' +res' if a.res else ''}: {ms*1e3:.1f} us, {fl/ms/1e9:.0f} TFLOP/s of conv work, {3*fl/ms/1e9:.0f} of MFMA work  [{e.layer_kernel(li, a.batch)}]")
if prof:
    names = ["loop top / item set-up", "H half-chunks (not last)", "L half-chunks (not last)", "last half-chunk (L)", "epilogue: vmcnt(0) drain", "epilogue: compute + stores"]
    for gq in (0, 1):
        n = out[7 + 16 * gq] / 5.0
        tot = sum(out[i + 16 * gq] for i in range(16) if i != 7) / 5.0
        print(f" wave group {gq}: {n:.0f} items per launch; mean cycles per item (wave 0 of the group):")
        for i, nm in enumerate(names):
            print(f"  {nm:34s} {out[i + 16 * gq]/5.0/max(n,1):9.0f}  {100*out[i + 16 * gq]/5.0/max(tot,1):5.1f}%")
        print(f"  {'total':34s} {tot/max(n,1):9.0f}")
e.close()
