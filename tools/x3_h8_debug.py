#!/usr/bin/env python3
"""Scratch: where does conv_h8x3 differ from torch?  Error by channel tile, by pixel-in-tile, by image row."""
import importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_pkg
load_pkg()
M = importlib.import_module("adas_amd.models"); CE = importlib.import_module("adas_amd.coreEngine")

def case(H, W, cin, cout, act, res, batch):
    ws = M.SynthWeights(0, gain=1.0)
    g = M.Graph("unit", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    r = a if (res and cin == cout) else None
    y = g.conv(a, cout, 3, 1, "test", act=act, res=r, res_mode=M.RES_BEFORE_ACT if r is not None else M.RES_NONE)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), "x3dbg.hipm"); g.save(path)
    e = CE.HipEngine(path, "fp16x3", batch)
    xin = np.random.default_rng(0).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("test", batch)
    kn = e.layer_kernel(e.layer_index("test"), batch)
    e.close()
    Wt = {k: torch.from_numpy(v) for k, v in ws.store.items()}
    with torch.no_grad():
        t = torch.from_numpy(xin)
        a_ = F.silu(F.conv2d(t, Wt["expand.weight"], Wt["expand.bias"]))
        yv = F.conv2d(a_, Wt["test.weight"], Wt["test.bias"], padding=1)
        if r is not None: yv = yv + a_
        yv = {M.ACT_NONE: lambda v: v, M.ACT_RELU: F.relu, M.ACT_SILU: F.silu}[act](yv)
    want = yv.numpy()
    d = np.abs(got - want)
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    print(f"== {H}x{W} {cin}->{cout} act {act} res {res} batch {batch}: {kn} rel {rel:.3e}")
    if rel > 1e-5:
        print("  err by 16-ch tile:", np.round(d.mean(axis=(0, 2, 3)).reshape(-1, 16).mean(1), 4))
        print("  err by channel%8 :", np.round(np.stack([d[:, k::8].mean() for k in range(8)]), 4))
        print("  err by image     :", np.round(d.mean(axis=(1, 2, 3))[:8], 4))
        print("  err by row (img0):", np.round(d[0].mean(axis=(0, 2))[:12], 4))
        print("  err by col (img0):", np.round(d[0].mean(axis=(0, 1))[:20], 4))
        print("  ref mean |v|     :", float(np.abs(want).mean()))
        # is it a permutation / partial sum?  correlate got with want
        print("  corr(got, want)  :", float(np.corrcoef(got.ravel(), want.ravel())[0, 1]))
        # try: only first 32 input channels
        with torch.no_grad():
            for nm, sl in (("first32", slice(0, 32)), ("last32", slice(cin - 32, cin))):
                yp = F.conv2d(a_[:, sl], Wt["test.weight"][:, sl], Wt["test.bias"], padding=1)
                print(f"  rel vs partial {nm}:", float(np.linalg.norm(got - yp.numpy()) / np.linalg.norm(yp.numpy())))

case(80, 400, 64, 64, M.ACT_NONE, False, 2)
case(80, 400, 64, 64, M.ACT_RELU, True, 2)
case(80, 400, 32, 64, M.ACT_NONE, False, 2)
case(40, 200, 128, 128, M.ACT_NONE, False, 8)
