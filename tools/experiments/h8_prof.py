#!/usr/bin/env python3
"""scratch: per-phase cycle breakdown of conv_halo_kernel (needs a build with ADAS_CFLAGS=-DADAS_H8_PROF).
   python tools/experiments/halo_prof.py --hw 80 400 --cin 64 --cout 64 --batch 64"""
import argparse, ctypes as C, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
M = importlib.import_module("adas_amd.models"); CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, nargs=2, default=[80, 400]); ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=64); ap.add_argument("--s", type=int, default=1); ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
H, W = a.hw
ws = M.SynthWeights(0, gain=1.0)
g = M.Graph("unit", 3, H * a.s, W * a.s, ws)
x, c3 = g.input()
e1 = g.conv(x, a.cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
y = g.conv(e1, a.cout, 3, a.s, "test", act=M.ACT_RELU)
z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
g.output(z, 0, [1, z.h * z.w * 8], "o")
path = os.path.join(tempfile.gettempdir(), "halo_prof.hipm"); g.save(path)
e = CE.HipEngine(path, "fp16", a.batch)
xin = np.random.default_rng(0).uniform(0, 1, (a.batch, 3, H * a.s, W * a.s)).astype(np.float32)
buf = L.DeviceBuffer.from_array(xin)
lib = C.CDLL(L.LIB_PATH)
out = (C.c_ulonglong * 32)()
e.profile(buf.ptr, a.batch, 2)
lib.adas_debug_h8_prof(out, 1)
rows = e.profile(buf.ptr, a.batch, 5)
lib.adas_debug_h8_prof(out, 0)
ms = [r[3] for r in rows if r[0] == "test"][0]
names = ["item setup", "first chunk", "epilogue: address math", "epilogue: drain + compute + stores", "last chunk: tap 8", "loop top", "middle chunks (all)", None] + ["last chunk: tap %d" % t for t in range(8)]
print(f"layer {ms*1e3:.1f} us")
for g in (0, 1):
    n = out[7 + 16 * g]
    tot = sum(out[i + 16 * g] for i in range(16) if i != 7)
    print(f" wave group {g}: {n} items; mean cycles per item:")
    for i, nm in enumerate(names):
        if nm: print(f"  {nm:40s} {out[i + 16 * g]/max(n,1):9.0f}  {100*out[i + 16 * g]/max(tot,1):5.1f}%")
    print(f"  {'total':40s} {tot/max(n,1):9.0f}")
