#!/usr/bin/env python3
"""scratch (round 6): per-phase cycle breakdown of conv_stem_pool_x3_kernel (the exact mode's ResNet stem + max-pool launch).  Needs
ADAS_LIB=<a library built with ADAS_BUILD_TAG=sp3prof ADAS_CFLAGS=-DADAS_SP3_PROF python vehicle-cv-adas_amd/build.py>:
   ADAS_LIB=vehicle-cv-adas_amd/_scratch/libadas_hip_sp3prof.so python tools/experiments/stem_pool_prof.py [--batch 64]"""
import argparse, ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
import netutil
CE = importlib.import_module("adas_amd.coreEngine"); L = CE.L
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
path, _, _ = netutil.model("ufldv2_res18")
e = CE.HipEngine(path, "fp16x3", a.batch)
xin = np.random.default_rng(0).uniform(-2, 2, (a.batch, 3, 320, 1600)).astype(np.float32)
buf = L.DeviceBuffer.from_array(xin)
lib = C.CDLL(L.LIB_PATH)
prof = hasattr(lib, "adas_debug_sp3_prof")
out = (C.c_ulonglong * 32)()
e.profile(buf.ptr, a.batch, 2)
if prof: lib.adas_debug_sp3_prof(out, 1)
rows = e.profile(buf.ptr, a.batch, 5)
if prof: lib.adas_debug_sp3_prof(out, 0)
li = [i for i, r in enumerate(rows) if r[0] == "model.conv1"][0]
print(f"model.conv1 batch {a.batch}: {rows[li][3]*1e3:.1f} us  [{e.layer_kernel(li, a.batch)}]")
if prof:
    names = ["barrier A (previous pool done)", "split + window ds_write + barrier B", "issue next tile's loads", "MFMAs (7 rows x 3 products)",
             "barrier C (+ bias loads)", "conv tile ds_write + barrier D", "pool + stores"]
    for gq, wv in ((0, 0), (1, 3)):
        n = out[7 + 16 * gq] / 5.0
        tot = sum(out[i + 16 * gq] for i in range(7)) / 5.0
        print(f" wave {wv}: {n/256:.1f} tiles per workgroup per launch; mean cycles per tile:")
        for i, nm in enumerate(names):
            print(f"  {nm:40s} {out[i + 16 * gq]/5.0/max(n,1):9.0f}  {100*out[i + 16 * gq]/5.0/max(tot,1):5.1f}%")
        print(f"  {'total':40s} {tot/max(n,1):9.0f}")
e.close()
