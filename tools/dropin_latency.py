#!/usr/bin/env python3
"""Per-call latency of the drop-in task classes on ONE host frame (the reference's calling pattern, demo.py:268-281): YoloDetector.DetectFrame,
BYTETracker.update, UltrafastLaneDetectorV2.DetectFrame (+ device lane geometry) -- upload, network, post-processing and the fetch of the
results, per frame, seeded random-weight models, fp16.   python tools/dropin_latency.py [--frames 200]"""
import argparse, importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
D = importlib.import_module("vehicle-cv-adas_amd.detectors")
A = importlib.import_module("vehicle-cv-adas_amd.analysis")
M = importlib.import_module("vehicle-cv-adas_amd.models")
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=200); a = ap.parse_args()
work = tempfile.mkdtemp(prefix="adas_lat_")
det_path = M.build("yolov8n").save(os.path.join(work, "yolov8n.hipm"))
lane_path = M.build("ufldv2_res18", wsrc=M.SynthWeights(1, gain=M.RELU_RES_GAIN)).save(os.path.join(work, "culane_res18.hipm"))
classes = os.path.join(work, "labels.txt"); open(classes, "w").write("\n".join("class%d" % i for i in range(80)) + "\n")
lane = D.UltrafastLaneDetectorV2(lane_path, D.LaneModelType.UFLDV2_CULANE)
det = D.YoloDetector(model_path=det_path, model_type=D.ObjectModelType.YOLOV8, classes_path=classes, box_score=0.4, box_nms_iou=0.45)
trk = D.BYTETracker()
frames = bench.cam_frames(4, 77)                      # (4, 720, 1280, 3) uint8, the bench's synthetic camera frames
lane.enable_device_geometry(A.PerspectiveTransformation((1280, 720)))
t = {"detect": [], "track": [], "lane": []}
for k in range(a.frames + 20):
    f = frames[k % 4]
    t0 = time.perf_counter(); det.DetectFrame(f)
    t1 = time.perf_counter()
    objs = det.object_info
    trk.update([o.tolist(format_type="xyxy") for o in objs], [o.conf for o in objs], [o.label for o in objs], f)
    t2 = time.perf_counter(); lane.DetectFrame(f)
    t3 = time.perf_counter()
    if k >= 20:
        t["detect"].append(t1 - t0); t["track"].append(t2 - t1); t["lane"].append(t3 - t2)
tot = sum(np.median(v) for v in t.values())
print("drop-in classes, one 1280x720 host frame per call, fp16, %d frames (median / p90, ms):" % a.frames)
for k, v in t.items():
    print("  %-7s %.3f / %.3f" % (k, 1e3 * np.median(v), 1e3 * np.percentile(v, 90)))
print("  sum of medians %.3f ms per frame = %.0f frames/s through the reference's own per-frame loop; %d objects on the last frame" % (1e3 * tot, 1.0 / tot, len(det.object_info)))
det.close(); lane.close(); trk.close()
