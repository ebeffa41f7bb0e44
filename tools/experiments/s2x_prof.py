#!/usr/bin/env python3
"""scratch (round 6): per-phase cycle breakdown of conv_s2p_x3_kernel (the split precision's 3x3 stride-2 conv).  Needs
ADAS_LIB=<a library built with ADAS_BUILD_TAG=s2xprof ADAS_CFLAGS=-DADAS_S2X_PROF python vehicle-cv-adas_amd/build.py>:
   ADAS_LIB=vehicle-cv-adas_amd/_scratch/libadas_hip_s2xprof.so python tools/experiments/s2x_prof.py --hw 80 400 --cin 64 --cout 128 --batch 64"""
import argparse, ctypes as C, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
M = importlib.import_module("adas_amd.models"); CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, nargs=2, default=[80, 400]); ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=128); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--act", default="relu")
a = ap.parse_args()
H, W = a.hw
ws = M.SynthWeights(0, gain=1.0)
g = M.Graph("unit", 3, H, W, ws)
x, c3 = g.input()
e1 = g.conv(x, a.cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
y = g.conv(e1, a.cout, 3, 2, "test", act=M.ACT_RELU if a.act == "relu" else M.ACT_SILU)
z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
g.output(z, 0, [1, z.h * z.w * 8], "o")
path = os.path.join(tempfile.gettempdir(), "s2x_prof.hipm"); g.save(path)
e = CE.HipEngine(path, "fp16x3", a.batch)
xin = np.random.default_rng(0).uniform(0, 1, (a.batch, 3, H, W)).astype(np.float32)
buf = L.DeviceBuffer.from_array(xin)
lib = C.CDLL(L.LIB_PATH)
prof = hasattr(lib, "adas_debug_s2x_prof")
dprof = hasattr(lib, "adas_debug_s2d_prof") and os.environ.get("ADAS_NO_S2D_X3") != "1"
if dprof: prof = False
out = (C.c_ulonglong * 16)()
e.profile(buf.ptr, a.batch, 2)
if prof: lib.adas_debug_s2x_prof(None, 1)
if dprof: lib.adas_debug_s2d_prof(None, 1)
rows = e.profile(buf.ptr, a.batch, 5)
if prof: lib.adas_debug_s2x_prof(out, 0)
if dprof: lib.adas_debug_s2d_prof(out, 0)
li = [i for i, r in enumerate(rows) if r[0] == "test"][0]
ms = rows[li][3]
fl = 2.0 * a.batch * (H // 2) * (W // 2) * a.cout * 9 * a.cin
print(f"{H}x{W}x{a.cin}->{a.cout} s2 batch {a.batch}: {ms*1e3:.1f} us, {fl/ms/1e9:.0f} TFLOP/s of conv work, {3*fl/ms/1e9:.0f} of MFMA work  [{e.layer_kernel(li, a.batch)}]")
if prof:
    names = ["set-up (addresses, bias)", "first half-chunk: loads -> LDS", "H: issue next loads + 9 taps", "H: barrier after the taps", "H: L window -> LDS + barrier",
             "L: issue next loads + 9 taps", "L: barrier + next H window/weights -> LDS + barrier", "epilogue"]
    n = out[8] / 5.0
    tot = sum(out[i] for i in range(8)) / 5.0
    print(f" {n:.0f} workgroups per launch ({n/256:.1f} per CU); mean cycles per workgroup (thread 0), {a.cin // 32} chunks:")
    for i, nm in enumerate(names):
        print(f"  {nm:52s} {out[i]/5.0/max(n,1):9.0f}  {100*out[i]/5.0/max(tot,1):5.1f}%")
    print(f"  {'total':52s} {tot/max(n,1):9.0f}")
if dprof:   # conv_s2d_x3_kernel (LDS-DMA form): thread 0's cycles per item
    names = ["prologue (first item's loads)", "H tap groups (MFMAs)", "L tap groups (MFMAs)", "counted vmcnt waits", "group barriers", "issue slots (DMA)", "epilogue", "next item's set-up"]
    n = out[8] / 5.0
    tot = sum(out[i] for i in range(8)) / 5.0
    print(f" {n:.0f} items per launch ({n/256:.1f} per workgroup); mean cycles per item (thread 0), {a.cin // 32} chunks:")
    for i, nm in enumerate(names):
        print(f"  {nm:40s} {out[i]/5.0/max(n,1):9.0f}  {100*out[i]/5.0/max(tot,1):5.1f}%")
    print(f"  {'total':40s} {tot/max(n,1):9.0f}")
e.close()
