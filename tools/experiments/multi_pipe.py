#!/usr/bin/env python3
"""scratch: P concurrent pipelines of S/P streams each (more independent graphs in flight) vs one pipeline of S streams."""
import importlib, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
bench.load_pkg()
L = importlib.import_module("adas_amd._lib"); M = importlib.import_module("adas_amd.models")
CE = importlib.import_module("adas_amd.coreEngine"); PL = importlib.import_module("adas_amd.pipeline")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wd = tempfile.mkdtemp()
fr = bench.det_frames(4, 10)
det_path, _, _ = bench.build_detector(M, CE, "yolov8n", fr, wd, "mp")
wl = M.SynthWeights(1, gain=M.RELU_RES_GAIN)
lane_path = M.build("ufldv2_res18", wsrc=wl).save(os.path.join(wd, "lane.hipm"))
for P in (1, 2, 4):
    s = S // P
    pipes = [PL.AdasPipeline(det_path, lane_path, n_streams=s, max_candidates=512) for _ in range(P)]
    dd = [L.DeviceBuffer.from_array(bench.det_frames(s, 20 + i)) for i in range(P)]
    dl = [L.DeviceBuffer.from_array(bench.lane_frames(s, 40 + i)) for i in range(P)]
    for _ in range(5):
        for p, a, b in zip(pipes, dd, dl): p.step(a.ptr, b.ptr)
    for p in pipes: p.sync()
    t0 = time.perf_counter()
    K = 40
    for _ in range(K):
        for p, a, b in zip(pipes, dd, dl): p.step(a.ptr, b.ptr)
    for p in pipes: p.sync()
    dt = time.perf_counter() - t0
    print(f"P={P} pipelines x {s} streams: {K*S/dt:.0f} fps, {dt/K*1e3:.3f} ms per {S} frames")
    for p in pipes: p.close()
