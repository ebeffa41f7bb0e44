#!/usr/bin/env python3
"""Per-call latency of the drop-in task classes on ONE host frame (the reference's calling pattern, demo.py:268-281): YoloDetector.DetectFrame,
BYTETracker.update, UltrafastLaneDetectorV2.DetectFrame (+ device lane geometry) -- upload, network, post-processing and the fetch of the
results, per frame.  Seeded random-weight models; the detector's class branch is CALIBRATED on the frames (bench.build_detector) so that
every frame carries detections and the tracker call is not the empty-frame path.  Both precisions: the exact mode (the classes' default)
and fp16; with and without handing the lane class the detector's staged frame (`lane.DetectFrame(det.staged_frame)`).

    python tools/dropin_latency.py [--frames 200]"""
import argparse, importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
bench.load_pkg()
D = importlib.import_module("adas_amd.detectors")
A = importlib.import_module("adas_amd.analysis")
M = importlib.import_module("adas_amd.models")
CE = importlib.import_module("adas_amd.coreEngine")
from oracle import preprocess

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=200); a = ap.parse_args()
work = tempfile.mkdtemp(prefix="adas_lat_")
frames = bench.cam_frames(4, 77)                      # (4, 720, 1280, 3) uint8, the bench's synthetic camera frames
seam = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for f in frames])
det_path, _, _ = bench.build_detector(M, CE, "yolov8n", seam, work, "lat", target_per_frame=30.0)
lane_path = M.build("ufldv2_res18", wsrc=M.SynthWeights(1, gain=M.RELU_RES_GAIN)).save(os.path.join(work, "culane_res18.hipm"))
classes = os.path.join(work, "labels.txt"); open(classes, "w").write("\n".join("class%d" % i for i in range(80)) + "\n")


def run(precision, share):
    lane = D.UltrafastLaneDetectorV2(lane_path, D.LaneModelType.UFLDV2_CULANE, precision=precision)
    det = D.YoloDetector(model_path=det_path, model_type=D.ObjectModelType.YOLOV8, classes_path=classes, box_score=0.4, box_nms_iou=0.45,
                         precision=precision)
    trk = D.BYTETracker()
    lane.enable_device_geometry(A.PerspectiveTransformation((1280, 720)))
    t = {"detect": [], "track": [], "lane": []}
    n_obj, n_trk = [], []
    for k in range(a.frames + 20):
        f = frames[(k // 4) % 4]                       # each scene held four frames: confirmed tracks, then new ones
        t0 = time.perf_counter(); det.DetectFrame(f)
        t1 = time.perf_counter()
        objs = det.object_info
        out = trk.update([o.tolist(format_type="xyxy") for o in objs], [o.conf for o in objs], [o.label for o in objs], f)
        t2 = time.perf_counter(); lane.DetectFrame(det.staged_frame if share else f)
        t3 = time.perf_counter()
        if k >= 20:
            t["detect"].append(t1 - t0); t["track"].append(t2 - t1); t["lane"].append(t3 - t2)
            n_obj.append(len(objs)); n_trk.append(len(out))
    tot = sum(np.median(v) for v in t.values())
    print("%-7s %-28s detect %.3f / %.3f   track %.3f / %.3f   lane %.3f / %.3f   sum %.3f ms = %5.0f frames/s   objects %.1f, tracked %.1f per frame" % (
        det.engine.precision, "lane reuses staged frame" if share else "each class uploads the frame",
        1e3 * np.median(t["detect"]), 1e3 * np.percentile(t["detect"], 90), 1e3 * np.median(t["track"]), 1e3 * np.percentile(t["track"], 90),
        1e3 * np.median(t["lane"]), 1e3 * np.percentile(t["lane"], 90), 1e3 * tot, 1.0 / tot, np.mean(n_obj), np.mean(n_trk)))
    det.close(); lane.close(); trk.close()


print("drop-in classes, one 1280x720 host frame per call, %d frames, median / p90 in ms (calibrated YOLOv8n + UFLDv2-R18 + ByteTrack):" % a.frames)
for precision in ("fp16", None):            # None: the classes' default = the exact mode
    for share in (False, True):
        run(precision, share)
