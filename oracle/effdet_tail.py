"""Oracle: the in-graph tail of an exported EfficientDet-D0 (anchors, box decode, clip, score threshold, per-class NMS).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference holds no EfficientDet graph, weights or golden
outputs -- it loads 'models/efficientdet-d0-coco_fp32.onnx' through onnxruntime (ObjectDetector/efficientdetDetector.py:38, :119)
and reads three outputs, boxes / class ids / confidences (:68-70).  This restates the published post-processing of the
architecture (Tan, Pang, Le, arXiv:1911.09070 section 4; the widely used PyTorch implementation's Anchors, BBoxTransform, ClipBoxes
and torchvision.ops.batched_nms steps):
  anchors   levels 3..7, strides 2^l; per cell 3 scales 2^(k/3) x 3 ratios (1.0,1.0), (1.4,0.7), (0.7,1.4); side = anchor_scale *
            stride * scale; centres stride/2 + i*stride; rows (level, y, x, scale, ratio); stored float32 as (y1, x1, y2, x2)
  decode    (dy, dx, dh, dw): centre = d * size_a + centre_a, size = exp(d) * size_a -> xyxy, clipped to [0, W-1] x [0, H-1]
  score     sigmoid(max class logit), class = first arg-max; candidate iff score > score_thr
  NMS       by descending score (ties: anchor order); suppress same-class boxes with IoU > iou_thr; at most max_det
Anchors, scores and boxes are float32 VALUES; every expression between them is evaluated in float64 (csrc/post_core.h
effdet_tail_frame follows the same convention, so the comparison does not hinge on a float32 libm)."""
import numpy as np

SCALES = (1.0, 1.2599210498948732, 1.5874010519681994)       # 2 ** (0/3), 2 ** (1/3), 2 ** (2/3)
RATIOS = ((1.0, 1.0), (1.4, 0.7), (0.7, 1.4))


def anchors(in_h, in_w, anchor_scale=4.0):
    """-> (A, 4) float32 (y1, x1, y2, x2), rows (level, y, x, scale, ratio)."""
    out = []
    for level in range(3, 8):
        stride = 2 ** level
        ys = np.arange(stride / 2, in_h, stride, dtype=np.float64)
        xs = np.arange(stride / 2, in_w, stride, dtype=np.float64)
        cy, cx = np.meshgrid(ys, xs, indexing="ij")
        per = []
        for sc in SCALES:
            for rx, ry in RATIOS:
                side = anchor_scale * stride * sc
                ax2, ay2 = side * rx / 2.0, side * ry / 2.0
                per.append(np.stack([cy - ay2, cx - ax2, cy + ay2, cx + ax2], -1))
        out.append(np.stack(per, 2).reshape(-1, 4))
    return np.concatenate(out, 0).astype(np.float32)


def tail(reg, cls, in_hw, score_thr=0.05, iou_thr=0.5, max_det=100, anchor_scale=4.0):
    """reg (A, 4), cls (A, nc) float32 of ONE frame -> dict(boxes (n,4) float32 xyxy, class_id (n,), conf (n,) float32, n_candidates)."""
    reg = np.asarray(reg, np.float32); cls = np.asarray(cls, np.float32)
    H, W = in_hw
    an = anchors(H, W, anchor_scale).astype(np.float64)
    cid = cls.argmax(1)
    best = cls[np.arange(len(cls)), cid].astype(np.float64)
    score = (1.0 / (1.0 + np.exp(-best))).astype(np.float32)
    cand = np.nonzero(score.astype(np.float64) > score_thr)[0]
    order = cand[np.argsort(-score[cand].astype(np.float64), kind="stable")]
    a, d = an[order], reg[order].astype(np.float64)
    ya, xa = (a[:, 0] + a[:, 2]) / 2.0, (a[:, 1] + a[:, 3]) / 2.0
    ha, wa = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    w, h = np.exp(d[:, 3]) * wa, np.exp(d[:, 2]) * ha
    yc, xc = d[:, 0] * ha + ya, d[:, 1] * wa + xa
    x1, y1 = np.maximum(xc - w / 2.0, 0.0), np.maximum(yc - h / 2.0, 0.0)
    x2, y2 = np.minimum(xc + w / 2.0, W - 1.0), np.minimum(yc + h / 2.0, H - 1.0)
    boxes = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
    b = boxes.astype(np.float64)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    c = cid[order]
    supp = np.zeros(len(order), bool)
    keep = []
    for i in range(len(order)):
        if supp[i]:
            continue
        keep.append(i)
        if len(keep) == max_det:
            break
        j = np.arange(i + 1, len(order))
        iw = np.minimum(b[i, 2], b[j, 2]) - np.maximum(b[i, 0], b[j, 0])
        ih = np.minimum(b[i, 3], b[j, 3]) - np.maximum(b[i, 1], b[j, 1])
        inter = np.maximum(iw, 0.0) * np.maximum(ih, 0.0)
        with np.errstate(invalid="ignore", divide="ignore"):
            iou = inter / (area[i] + area[j] - inter)
        supp[j] |= (c[j] == c[i]) & (iou > iou_thr)
    keep = np.asarray(keep, np.int64)
    return dict(boxes=boxes[keep], class_id=c[keep].astype(np.int64), conf=score[order][keep], n_candidates=int(len(cand)), anchor=order[keep])
