#!/usr/bin/env python3
"""Single conv layer microbenchmark (through a tiny graph): input(3) -> 1x1 expand -> conv under test.
    python tools/bench_conv.py --hw 80 400 --cin 64 --cout 64 --k 3 --s 1 --batch 16 --iters 50
Prints ms and algorithmic TFLOP/s of the layer under test (hipEvents per layer)."""
import argparse, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
M = importlib.import_module("adas_amd.models"); CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, nargs=2, default=[80, 400]); ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=64); ap.add_argument("--k", type=int, default=3); ap.add_argument("--s", type=int, default=1)
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--precision", default="bf16"); ap.add_argument("--act", type=int, default=2)
a = ap.parse_args()
H, W = a.hw
ws = M.SynthWeights(0, gain=1.0)
g = M.Graph("unit", 3, H * a.s, W * a.s, ws)       # the tested layer's OUTPUT is HxW
x, c3 = g.input()
e1 = g.conv(x, a.cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
y = g.conv(e1, a.cout, a.k, a.s, "test", act=a.act)
z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
g.output(z, 0, [1, z.h * z.w * 8], "o")
path = os.path.join(tempfile.gettempdir(), "bench_conv.hipm"); g.save(path)
e = CE.HipEngine(path, a.precision, a.batch)
xin = np.random.default_rng(0).uniform(0, 1, (a.batch, 3, H * a.s, W * a.s)).astype(np.float32)
buf = L.DeviceBuffer.from_array(xin)
e.profile(buf.ptr, a.batch, 3)
rows = e.profile(buf.ptr, a.batch, a.iters)
for name, fl, kind, ms in rows:
    if name == "test":
        print(f"conv {y.h}x{y.w} {a.cin}->{a.cout} k{a.k}s{a.s} batch {a.batch} {a.precision}: {ms*1e3:.1f} us, "
              f"{fl*a.batch/1e9:.2f} GFLOP, {fl*a.batch/(ms*1e-3)/1e12:.1f} TFLOP/s  (NO_HALO={os.environ.get('ADAS_NO_HALO','0')})")
