"""TEST INFRASTRUCTURE (oracle) -- NumPy restatement of the reference's EfficientDet wrapper.

  prepare_input   <- EfficientdetDetector.__prepare_input (ObjectDetector/efficientdetDetector.py:57-65)
  process_output  <- EfficientdetDetector.__process_output (:67-85) over Scaler.convert_boxes_coordinate (utils.py:70-87)

The exported EfficientDet graph (README.md model table, efficientdet-d0..d3) decodes and suppresses inside the ONNX file; its
outputs are boxes (n, 4) xyxy float32 in input pixels, class ids (n), confidences (n).  PINNED: tests/golden/effdet_post.npz holds
the outputs of the reference's own __process_output under the stubs of make_golden.py (tests/golden/make_golden_effdet.py).
Dtype notes: a float32 array combined with Python ints/floats stays float32 in NumPy 1.22 and 2.x alike, so the inverse letterbox
is float32 arithmetic; `conf < box_score` compares a float32 scalar with a Python float (double comparison in the pinned env).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np

from . import preprocess

MEAN = (0.406, 0.456, 0.485)     # BGR order: the reference does not swap channels here (:57)
STD = (0.225, 0.224, 0.229)


def prepare_input(srcimg_bgr, target_hw):
    """-> (1,3,H,W) float32."""
    canvas, _, _ = preprocess.letterbox_image(srcimg_bgr, target_hw)
    image = (canvas / 255 - MEAN) / STD
    return np.transpose(np.expand_dims(image, axis=0), (0, 3, 1, 2)).astype(np.float32)


def process_output(boxes, ids, confs, lb, box_score, n_classes=None):
    """lb: oracle.yolo_post.letterbox_params(...) dict (pad, ratio).  Returns dict(xywh float32 (k,4), conf float32 (k,), class_id (k,),
    xyxy_int (k,4) = RectInfo.tolist())."""
    boxes = np.array(boxes, dtype=np.float32).reshape(-1, 4).copy()
    ids = np.asarray(ids).reshape(-1)
    confs = np.asarray(confs, np.float32).reshape(-1)
    if boxes.size > 0:
        ratioh, ratiow = lb["ratio"]
        padh, padw = lb["pad"]
        boxes[..., [0, 2]] = (boxes[..., [0, 2]] - padw) * ratiow
        boxes[..., [1, 3]] = (boxes[..., [1, 3]] - padh) * ratioh
        boxes[:, 2:4] = boxes[:, 2:4] - boxes[:, 0:2]
    keep = [i for i in range(len(boxes)) if not (confs[i] < box_score)]
    xywh = boxes[keep].reshape(-1, 4)
    xyxy = np.stack([xywh[:, 0], xywh[:, 1], xywh[:, 0] + xywh[:, 2], xywh[:, 1] + xywh[:, 3]], axis=1) if len(keep) else np.zeros((0, 4), np.float32)
    return dict(xywh=xywh, conf=confs[keep], class_id=ids[keep].astype(np.int64), xyxy_int=xyxy.astype(np.float32).astype(np.int64))
