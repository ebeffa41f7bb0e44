"""Single-frame (batch 1) latency of the legacy coreEngine surface (engine_inference: H2D + forward + D2H) and of the
device path.  An engine-level hipGraph replay was tried here and measured identical (0.72 ms for YOLOv8n either way): at
batch 1 the ~100 kernels are bound by per-dispatch latency on the GPU side, not by host launch cost, so it was dropped."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import netutil
from conftest import load_pkg
load_pkg()
CE = importlib.import_module("adas_amd.coreEngine")
L = importlib.import_module("adas_amd._lib")

for name, shape in (("yolov8n", (1, 3, 640, 640)), ("ufldv2_res18", (1, 3, 320, 1600)), ("yolov8s", (1, 3, 640, 640))):
    path, W, g = netutil.model(name)
    x = np.random.default_rng(0).uniform(0, 1, shape).astype(np.float32)
    for mode in ("-",):
        e = CE.HipEngine(path, precision="bf16", max_batch=1)
        for _ in range(5):
            out = e.engine_inference(x)
        t0 = time.perf_counter()
        for _ in range(200):
            out = e.engine_inference(x)
        t1 = time.perf_counter()
        buf = L.DeviceBuffer.from_array(x)
        for _ in range(5):
            e.infer_device(buf.ptr, 1, None)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(200):
            e.infer_device(buf.ptr, 1, None)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("%-14s graph=%s  engine_inference %.3f ms/frame   infer_device %.3f ms/frame  checksum %.6f" % (
            name, mode, (t1 - t0) * 5, (t3 - t2) * 5, float(np.abs(out[0]).sum())))
        buf.free(); e.close()
