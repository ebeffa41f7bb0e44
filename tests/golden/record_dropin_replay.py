#!/usr/bin/env python3
"""Record the HipEngine surface + one output set per detector, with the device results that belong to them (GPU box).

    python tests/golden/record_dropin_replay.py [out.npz]      # default: gpurun_out/dropin_replay.npz on the GPU box

INTEGRATION.md section A says: swap the reference's `coreEngine` module for this repo's and the reference's own detector classes keep
working.  The GPU box has no reference tree and the build container has no GPU, so the claim is tested in two halves that meet in this
fixture:

  here (GPU)    this repo's drop-in YoloDetector / UltrafastLaneDetectorV2 process one 1280x720 frame in the default (exact) precision;
                recorded: what the reference's classes read from an engine (framework_type, providers, engine_dtype, input shape, output
                shapes + names), the tensors `engine_inference` returned for that frame, a digest of the input tensor, and the device's
                `object_info` / `lane_info`
  there (CPU)   tests/test_integration_replay.py imports the REFERENCE's unmodified YoloDetector / UltrafastLaneDetectorV2 with
                `coreEngine` replaced by a replay engine exposing exactly the recorded surface, feeds the same frame, and compares
                their `object_info` / `lane_info` with the device's

The fixture is DATA (a frame, tensors, results); no reference code is involved in making it.  Detector: YOLOv8n at 640x640 with 16
classes (a (1, 20, 8400) head: 0.7 MB instead of 2.8), class branch calibrated like bench.py's; lane net: UFLDv2-R18, CULane geometry.
"""
import hashlib
import importlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

NC = 16
LABELS = ["label%02d" % i for i in range(NC - 1)]      # one short: the last class id maps to "unknown" (yoloDetector.py:144-147)


def replay_frame(seed, h=720, w=1280):
    """A BGR u8 camera-like frame that compresses well: an 8x8-block colour field with filled rectangles (no per-pixel noise)."""
    rng = np.random.default_rng(seed)
    img = np.repeat(np.repeat(rng.integers(0, 255, ((h + 7) // 8, (w + 7) // 8, 3)), 8, 0), 8, 1)[:h, :w]
    for _ in range(24):
        x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
        x1, y1 = min(w, x0 + int(rng.integers(30, 400))), min(h, y0 + int(rng.integers(30, 300)))
        img[y0:y1, x0:x1] = rng.integers(0, 255, 3)
    return np.ascontiguousarray(img.astype(np.uint8))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    import bench
    bench.load_pkg()
    L = importlib.import_module("adas_amd._lib")
    M = importlib.import_module("adas_amd.models")
    CE = importlib.import_module("adas_amd.coreEngine")
    D = importlib.import_module("adas_amd.detectors")
    from oracle import preprocess
    if L.lib().adas_device_count() <= 0:
        raise SystemExit("record_dropin_replay.py needs an MI355X")
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "dropin_replay.npz")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    work = tempfile.mkdtemp(prefix="adas_replay_")
    lab = os.path.join(work, "labels.txt")
    open(lab, "w").write("\n".join(LABELS))

    # lane net first: pick the first frame seed whose ego lanes are both found (random weights find them on about half the frames)
    wl = M.SynthWeights(1, gain=M.RELU_RES_GAIN)
    lane_path = M.build("ufldv2_res18", wsrc=wl).save(os.path.join(work, "lane.hipm"))
    lane = D.UltrafastLaneDetectorV2(lane_path, D.LaneModelType.UFLDV2_CULANE)
    best = None
    for cand in range(100, 356):
        lane.DetectFrame(replay_frame(cand))
        li = lane.lane_info
        n_found = sum(bool(s) for s in li.lanes_status)
        key = (bool(li.area_status), n_found, sum(len(p) for p in li.lanes_points if p is not None))
        if best is None or key > best[0]:
            best = (key, cand)
        if key[0] and n_found >= 3:
            break
    (has_area, n_found, n_points), seed = best
    print("frame seed %d: ego-lane area %s, %d lanes found, %d lane points (best of the candidates examined)" % (seed, has_area, n_found, n_points))
    if n_found < 1:
        raise SystemExit("no candidate frame carries a detected lane")
    frame = replay_frame(seed)

    # detector: calibrated on this frame and three neighbours (about 40 anchors over box_score on the median frame)
    cal = np.concatenate([preprocess.yolo_prepare_input(replay_frame(s), (640, 640)) for s in (seed, seed + 1000, seed + 2000, seed + 3000)])
    det_path, _, _ = bench.build_detector(M, CE, "yolov8n", cal, work, "replay", target_per_frame=40.0, build_kw=dict(nc=NC))
    det = D.YoloDetector(model_path=det_path, model_type=D.ObjectModelType.YOLOV8, classes_path=lab, box_score=0.4, box_nms_iou=0.45)

    rec = {"frame": frame, "frame_seed": np.int64(seed), "labels": np.array(LABELS)}
    # ---- detector
    det.DetectFrame(frame)
    x = det._stage.tensor.download((1, 3, 640, 640), np.float32)
    np.testing.assert_array_equal(x, preprocess.yolo_prepare_input(frame, (640, 640)))      # the device tensor IS the reference's blob
    outs = det.engine.engine_inference(x)
    shapes, names = det.engine.get_engine_output_shape()
    info = det.object_info
    assert len(info) >= 3, "calibration gave %d detections" % len(info)
    rec.update({
        "det_precision": np.array(det.engine.precision), "det_framework_type": np.array(det.engine.framework_type),
        "det_providers": np.array(det.engine.providers), "det_engine_dtype": np.array(np.dtype(det.engine.engine_dtype).name),
        "det_input_shape": np.asarray(det.engine.get_engine_input_shape(), np.int64),
        "det_output_shapes": np.asarray(shapes, np.int64), "det_output_names": np.array(names),
        "det_input_sha256": np.array(digest(x)), "det_out0": outs[0],
        "det_xywh": np.asarray([[r.x, r.y, r.width, r.height] for r in info], np.float64).reshape(-1, 4),
        "det_conf": np.asarray([r.conf for r in info], np.float64), "det_label": np.array([r.label for r in info]),
        "det_xyxy_int": np.asarray([r.tolist() for r in info], np.int64).reshape(-1, 4),
    })
    # ---- lane
    lane.DetectFrame(frame)
    xl = lane._stage.tensor.download((1, 3, 320, 1600), np.float32)
    np.testing.assert_array_equal(xl, preprocess.ufld_prepare_input(frame, (320, 1600), 0.6))
    louts = lane.engine.engine_inference(xl)
    lshapes, lnames = lane.engine.get_engine_output_shape()
    li = lane.lane_info
    pts = [np.asarray(p if p is not None else [], np.int64).reshape(-1, 2) for p in li.lanes_points]
    rec.update({
        "lane_precision": np.array(lane.engine.precision), "lane_framework_type": np.array(lane.engine.framework_type),
        "lane_providers": np.array(lane.engine.providers), "lane_engine_dtype": np.array(np.dtype(lane.engine.engine_dtype).name),
        "lane_input_shape": np.asarray(lane.engine.get_engine_input_shape(), np.int64),
        "lane_output_ndims": np.asarray([len(s) for s in lshapes], np.int64),
        "lane_output_shapes": np.asarray([list(s) + [0] * (4 - len(s)) for s in lshapes], np.int64), "lane_output_names": np.array(lnames),
        "lane_input_sha256": np.array(digest(xl)),
        "lane_status": np.asarray([bool(s) for s in li.lanes_status]), "lane_area_status": np.bool_(li.area_status),
        "lane_area_points": np.asarray(li.area_points, np.int64).reshape(-1, 2),
    })
    for i in range(4):
        rec["lane_out%d" % i] = louts[i]
        rec["lane_points%d" % i] = pts[i]
    np.savez_compressed(out_path, **rec)
    print("wrote %s (%.0f KB): frame seed %d, %d detections (%s), lanes %s, area %s with %d points, precision %s / %s" % (
        out_path, os.path.getsize(out_path) / 1024, seed, len(info), sorted(set(r.label for r in info)), [len(p) for p in pts],
        bool(li.area_status), len(rec["lane_area_points"]), det.engine.precision, lane.engine.precision))
    det.close(); lane.close()


if __name__ == "__main__":
    main()
