#!/usr/bin/env python3
"""Per-layer device time (hipEvents) of a model at a given batch: ms, GFLOP, TFLOP/s, tile.
    python tools/profile_layers.py yolov8n --batch 16 [--precision bf16] [--top 25]"""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import netutil

ap = argparse.ArgumentParser()
ap.add_argument("model"); ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--precision", default="bf16"); ap.add_argument("--top", type=int, default=30)
ap.add_argument("--json", default=None)
a = ap.parse_args()
CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
path, W, g = netutil.model(a.model)
e = CE.HipEngine(path, a.precision, a.batch)
shp = e.get_engine_input_shape()
x = np.random.default_rng(0).uniform(0, 1, [a.batch] + shp[1:]).astype(np.float32)
buf = L.DeviceBuffer.from_array(x)
e.profile(buf.ptr, a.batch, 2)
rows = e.profile(buf.ptr, a.batch, 10)
ops = {o["name"]: o for o in g.ops}
tot = sum(r[3] for r in rows); totf = sum(r[1] for r in rows) * a.batch
print(f"{a.model} batch {a.batch} {a.precision}: {tot:.3f} ms/step, {totf/1e9:.1f} GFLOP/step, {totf/tot/1e9:.1f} TFLOP/s overall")
out = []
kern = {r[0]: e.layer_kernel(i, a.batch) for i, r in enumerate(rows)}
for name, fl, kind, ms in rows:
    o = ops.get(name)
    desc = ""
    if o is not None and o["type"] in (1, 7):
        v = o["ins"][0]; ov = o["out"]
        desc = f"{v.h}x{v.w}x{v.c}->{ov.c} k{o['kh']}s{o['stride']}"
    desc = f"{desc:28s} {kern[name]}"
    out.append(dict(name=name, ms=ms, gflop=fl * a.batch / 1e9, tflops=(fl * a.batch / (ms * 1e-3) / 1e12 if ms > 0 else 0), desc=desc, kind=kind))
for r in sorted(out, key=lambda r: -r["ms"])[:a.top]:
    print(f"{r['ms']:8.4f} ms {100*r['ms']/tot:5.1f}%  {r['gflop']:8.2f} GF {r['tflops']:7.1f} TF/s  {r['name']:28s} {r['desc']}")
if a.json:
    json.dump(out, open(a.json, "w"))
