#!/usr/bin/env python3
"""scratch (round 5): the SAME 3x3 halo tiles inside the multi-layer launch and as per-layer launches, phase by phase
(needs ADAS_LIB=<library built with -DADAS_HALO_PROF -DADAS_ML_PROF>).
   python tools/ml_hprof.py --hw 40 40 --c 64 --batch 64"""
import argparse, ctypes as C, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
M = importlib.import_module("adas_amd.models"); CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, nargs=2, default=[40, 40]); ap.add_argument("--c", type=int, default=64); ap.add_argument("--cout", type=int, default=0)
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--layers", type=int, default=2)
a = ap.parse_args()
H, W = a.hw
cout = a.cout or a.c
g = M.Graph("unit", 3, H, W, M.SynthWeights(0, gain=1.0))
x, c3 = g.input()
y = g.conv(x, a.c, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
for i in range(a.layers):
    y = g.conv(y, cout, 3, 1, "t%d" % i, act=M.ACT_SILU)
z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
g.output(z, 0, [1, z.h * z.w * 8], "o")
path = os.path.join(tempfile.gettempdir(), "ml_hprof.hipm"); g.save(path)
xin = np.random.default_rng(0).uniform(0, 1, (a.batch, 3, H, W)).astype(np.float32)
buf = L.DeviceBuffer.from_array(xin)
lib = C.CDLL(L.LIB_PATH)
names = ["setup", "issue first loads", "first loads land + LDS store", "barrier", "prefetch issue", "tap loop", "barrier (reads done)",
         "LDS store (+wait loads)", "barrier", "epilogue"]
res = {}
for ml in (1, 0):
    os.environ["ADAS_ML"] = "1" if ml else "0"
    os.environ["ADAS_NO_GROUP"] = "1"
    os.environ["ADAS_ML_ONLY"] = "halo"
    e = CE.HipEngine(path, "fp16", a.batch)
    e.prepare(a.batch)
    fn = lib.adas_debug_ml_halo_prof if ml else lib.adas_debug_halo_prof
    out = (C.c_ulonglong * 16)()
    e.profile(buf.ptr, a.batch, 2)
    fn(out, 1)
    rows = e.profile(buf.ptr, a.batch, 5)
    fn(out, 0)
    ms = sum(r[3] for r in rows if r[0].startswith("t"))
    kern = [e.layer_kernel(i, a.batch) for i in range(e.stats()["num_layers"])]
    res[ml] = (ms, list(out), kern)
    e.close()
for ml in (0, 1):
    ms, out, kern = res[ml]
    n = out[15]
    tot = sum(out[:10])
    print(("multi-layer launch" if ml else "per-layer launches"), "%.1f us for the %d layers; %d tile samples; kernels %s" % (ms * 1e3, a.layers, n, [k for k in kern if "halo" in k or "ml" in k]))
    for i, nm in enumerate(names):
        print("  %-32s %9.0f  %5.1f%%" % (nm, out[i] / max(n, 1), 100 * out[i] / max(tot, 1)))
    print("  %-32s %9.0f" % ("total", tot / max(n, 1)))
