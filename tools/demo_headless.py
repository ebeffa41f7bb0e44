#!/usr/bin/env python3
"""Headless counterpart of the reference's demo.py main loop (demo.py:230-296) on the drop-in classes: per frame
detector -> tracker, lane detector (+ device geometry), distance / collision point, FCWS / LDWS / LKAS state machine.
No window, no video codec: frames come from a seeded synthetic 1280x720 clip (moving rectangles on noise) or from a
.npy file of uint8 BGR frames (N, H, W, 3); one summary line per frame.

    python tools/demo_headless.py [--frames 30] [--det yolov8n.onnx|.hipm] [--det-type yolov8|efficientdet] [--lane culane_res18.onnx|.hipm] [--video clip.npy]

Without model paths, seeded random-weight models are built (their detections are noise: the point is the data flow)."""
import argparse, importlib, os, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

D = importlib.import_module("vehicle-cv-adas_amd.detectors")
A = importlib.import_module("vehicle-cv-adas_amd.analysis")
M = importlib.import_module("vehicle-cv-adas_amd.models")


def synthetic_clip(n, h=720, w=1280, seed=3):
    """SURVEY 8d C4 recipe: rectangles with constant velocity + jitter + 10 % dropout on grey noise."""
    rng = np.random.default_rng(seed)
    k = int(rng.integers(8, 30))
    pos = rng.uniform([100, 100], [w - 100, h - 100], (k, 2)); vel = rng.normal(0, 5, (k, 2))
    size = rng.uniform(40, 200, (k, 2)); col = rng.integers(0, 255, (k, 3))
    for _ in range(n):
        img = rng.normal(114, 20, (h, w, 3)).clip(0, 255).astype(np.uint8)
        pos += vel + rng.normal(0, 1, (k, 2))
        for i in range(k):
            if rng.uniform() < 0.1:
                continue
            x0, y0 = (pos[i] - size[i] / 2).astype(int); x1, y1 = (pos[i] + size[i] / 2).astype(int)
            img[max(0, y0):max(0, y1), max(0, x0):max(0, x1)] = col[i]
        yield img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--det", default=None); ap.add_argument("--lane", default=None); ap.add_argument("--video", default=None)
    ap.add_argument("--classes", default=None, help="label file, one name per line (default: 80 generic names)")
    ap.add_argument("--det-type", default="yolov8", choices=["yolov8", "efficientdet"],
                    help="demo.py picks EfficientdetDetector for ObjectModelType.EfficientDet and YoloDetector otherwise")
    a = ap.parse_args()
    work = tempfile.mkdtemp(prefix="adas_demo_")
    effdet = a.det_type == "efficientdet"
    det_path = a.det or (M.build("efficientdet-d0").save(os.path.join(work, "efficientdet-d0.hipm")) if effdet
                         else M.build("yolov8n").save(os.path.join(work, "yolov8n.hipm")))
    lane_path = a.lane or M.build("ufldv2_res18", wsrc=M.SynthWeights(1, gain=M.RELU_RES_GAIN)).save(os.path.join(work, "culane_res18.hipm"))
    classes = a.classes
    if classes is None:
        classes = os.path.join(work, "labels.txt")
        names = ["person", "bicycle", "car", "motorbike", "aeroplane", "bus", "train", "truck"] + ["class%d" % i for i in range(8, 90 if effdet else 80)]
        open(classes, "w").write("\n".join(names) + "\n")

    # demo.py:236-260
    laneDetector = D.UltrafastLaneDetectorV2(lane_path, D.LaneModelType.UFLDV2_CULANE)
    if effdet:     # the EfficientDet-D0 network + its in-graph decode / NMS on the device (coreEngine.EfficientdetEngine)
        objectDetector = D.EfficientdetDetector(model_path=det_path, classes_path=classes, box_score=0.1 if a.det is None else 0.6)
    else:
        objectDetector = D.YoloDetector(model_path=det_path, model_type=D.ObjectModelType.YOLOV8, classes_path=classes, box_score=0.4, box_nms_iou=0.45)
    frames = np.load(a.video) if a.video else None
    first = frames[0] if frames is not None else next(synthetic_clip(1))
    height, width = first.shape[:2]
    transformView = A.PerspectiveTransformation((width, height))
    laneDetector.enable_device_geometry(transformView)
    distanceDetector = A.SingleCamDistanceMeasure()
    objectTracker = D.BYTETracker()
    analyzeMsg = A.TaskConditions()

    src = iter(frames) if frames is not None else synthetic_clip(a.frames)
    t0 = time.perf_counter()
    n = 0
    for frame in src:
        if n >= a.frames:
            break
        # demo.py:268-281
        objectDetector.DetectFrame(frame)
        box = [obj.tolist(format_type="xyxy") for obj in objectDetector.object_info]
        score = [obj.conf for obj in objectDetector.object_info]
        ids = [obj.label for obj in objectDetector.object_info]
        tracks = objectTracker.update(box, score, ids, frame)
        laneDetector.DetectFrame(frame)
        # demo.py:284-296
        distanceDetector.updateDistance(objectDetector.object_info)
        vehicle_distance = distanceDetector.calcCollisionPoint(laneDetector.lane_info.area_points)
        if analyzeMsg.CheckStatus() and laneDetector.lane_info.area_status:
            transformView.updateTransformParams(*laneDetector.lane_info.lanes_points[1:3], analyzeMsg.transform_status)
        (vehicle_direction, vehicle_curvature), vehicle_offset = laneDetector.curve_and_offset
        analyzeMsg.UpdateCollisionStatus(vehicle_distance, laneDetector.lane_info.area_status)
        analyzeMsg.UpdateOffsetStatus(vehicle_offset)
        analyzeMsg.UpdateRouteStatus(vehicle_direction, vehicle_curvature)
        print("frame %3d: %2d objects, %2d tracks | lanes %s area %-5s | dir %-4s R %s off %s | FCWS %-8s LDWS %-8s LKAS %s" % (
            n, len(box), len(tracks), "".join("1" if s else "0" for s in laneDetector.lane_info.lanes_status),
            laneDetector.lane_info.area_status, vehicle_direction,
            "%.0f m" % vehicle_curvature if vehicle_curvature is not None else "-",
            "%+.2f m" % vehicle_offset if vehicle_offset is not None else "-",
            analyzeMsg.collision_msg.name, analyzeMsg.offset_msg.name, analyzeMsg.curvature_msg.name))
        n += 1
    dt = time.perf_counter() - t0
    print("%d frames in %.2f s (%.1f frames/s, single stream, host frames: upload + 2 nets + post-processing + fetch per frame)" % (n, dt, n / dt))
    objectDetector.close(); laneDetector.close(); objectTracker.close()
    return n


if __name__ == "__main__":
    main()
