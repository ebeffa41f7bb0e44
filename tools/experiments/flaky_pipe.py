#!/usr/bin/env python3
"""scratch: repeat test_pipeline_step_from_camera_frames' comparison many times in one process and report the first mismatches."""
import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_pkg
load_pkg()
import netutil, bench
L = importlib.import_module("adas_amd._lib"); PL = importlib.import_module("adas_amd.pipeline"); PP = importlib.import_module("adas_amd.postproc")
M = importlib.import_module("adas_amd.models")
S = 2; reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cam = [bench.cam_frames(S, 70 + i) for i in range(2)]
lane_path, _, _ = netutil.model("ufldv2_res18")
det_path = M.build("yolov8n").save("/tmp/flaky_d.hipm")
pa = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="bf16", src_hw=(720, 1280), use_graph=True)
pb = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="bf16", src_hw=(720, 1280), use_graph=False)
dt = L.DeviceBuffer(S * 3 * 640 * 640 * 4); lt = L.DeviceBuffer(S * 3 * 320 * 1600 * 4)
bad = 0
for r in range(reps):
    k = r & 1
    dc = L.DeviceBuffer.from_array(cam[k])
    pa.step_frames(dc.ptr, (720, 1280), 0.6); pa.sync()
    L.check(L.lib().adas_preprocess_yolo(dc.ptr, S, 720, 1280, dt.ptr, 640, 640, 1, None))
    L.check(L.lib().adas_preprocess_ufld(dc.ptr, S, 720, 1280, lt.ptr, 320, 1600, C.c_double(0.6), None))
    L.check(L.lib().adas_synchronize())   # the stand-alone kernels ran on the null stream; the pipeline's streams are non-blocking
    pb.step(dt.ptr, lt.ptr); pb.sync()
    for s in range(S):
        a, b = PP.YoloPost.fetch(pa.post, s), PP.YoloPost.fetch(pb.post, s)
        for key in ("cand_anchor", "cand_conf", "keep", "xyxy_int"):
            if a[key].shape != b[key].shape or not np.array_equal(a[key], b[key]):
                bad += 1
                n = min(len(a[key]), len(b[key]))
                d = np.abs(np.asarray(a[key][:n], np.float64) - np.asarray(b[key][:n], np.float64))
                print(f"rep {r} stream {s} {key}: shapes {a[key].shape} {b[key].shape} max|diff| {d.max() if n else -1} ndiff {(d>0).sum()}", flush=True)
                break
        la, lb = pa.decode.fetch(s), pb.decode.fetch(s)
        if la != lb:
            bad += 1; print(f"rep {r} stream {s} lanes differ", flush=True)
    dc.free()
print("mismatching (rep, stream) pairs:", bad, "of", reps * S)
