"""Oracle: YOLO head decode -> inverse letterbox -> NMS -> RectInfo.

TEST INFRASTRUCTURE (see oracle/__init__.py).  NumPy restatement of
  ObjectDetector/yoloDetector.py:104-157   (__process_output, get_nms_results)
  ObjectDetector/utils.py:42-87            (Scaler geometry + convert_boxes_coordinate)
  ObjectDetector/utils.py:105-159          (NMS.fast_nms, the alternative)
  ObjectDetector/utils.py:161-256          (NMS.fast_soft_nms, the production call)
  ObjectDetector/core.py:8-23              (RectInfo.tolist)

Numeric contract = the reference's *pinned* environment (numpy==1.22.1,
legacy value-based promotion): from yoloDetector.py:132 onward everything is
float64 arithmetic on float32-exact inputs (SURVEY.md finding 5).
"""
import numpy as np

V5_FAMILY = ("yolov5", "yolov5_lite", "yolov6", "yolov7")
V8_FAMILY = ("yolov8", "yolov9", "yolov10")


# --------------------------------------------------------------------------
# utils.py:42-68  Scaler.process_image geometry (no pixels) + get_scale_ratio
# --------------------------------------------------------------------------
def letterbox_params(src_hw, target_hw, keep_ratio=True):
    """Returns dict(old=(H,W), new=(newh,neww), pad=(padh,padw), ratio=(rh,rw))."""
    H, W = int(src_hw[0]), int(src_hw[1])
    Ht, Wt = int(target_hw[0]), int(target_hw[1])
    padh, padw, newh, neww = 0, 0, Ht, Wt
    if keep_ratio and H != W:
        hw_scale = H / W
        if hw_scale > 1:
            newh, neww = Ht, int(Wt / hw_scale)
            padw = int((Wt - neww) * 0.5)
        else:
            newh, neww = int(Ht * hw_scale) + 1, Wt
            padh = int((Ht - newh) * 0.5)
    return dict(old=(H, W), new=(newh, neww), pad=(padh, padw),
                ratio=(H / newh, W / neww))


# --------------------------------------------------------------------------
# yoloDetector.py:18-49  YoloLiteParameters.lite_postprocess (YOLOv5-lite heads)
# --------------------------------------------------------------------------
LITE_ANCHORS = np.asarray([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]],
                          np.float32).reshape(3, 3, 2)                      # :23,29
LITE_STRIDES = (8, 16, 32)                                                  # :28


def lite_postprocess(outs, input_hw):
    """Grid decode of a raw v5-lite head (A, 5+nc) float32 -> new array, fp32 arithmetic in the
    reference's order.  Level i holds na=3 blocks of h*w rows; the reference's grid is
    meshgrid(arange(h), arange(w)) flattened (:32-34 with the (w, h) call of :43), so cell n maps
    to (n % h, n // h) -- the textbook (n % w, n // w) only when h == w."""
    o = np.array(outs, dtype=np.float32, copy=True)
    row = 0
    two, half = np.float32(2.0), np.float32(0.5)
    for i, stride in enumerate(LITE_STRIDES):
        h, w = int(input_hw[0] / stride), int(input_hw[1] / stride)          # :40
        cells = h * w
        length = 3 * cells
        n = np.arange(cells)
        grid = np.stack([n % h, n // h], 1).astype(np.float32)             # :32-34
        sl = slice(row, row + length)
        blk = o[sl]
        if blk.shape[0] != length:
            raise ValueError("head has %d rows, level %d needs %d more" % (o.shape[0], i, length))
        xy = ((blk[:, 0:2] * two - half) + np.tile(grid, (3, 1))) * np.float32(stride)          # :45-46
        wh = ((blk[:, 2:4] * two) * (blk[:, 2:4] * two)) * np.repeat(LITE_ANCHORS[i], cells, axis=0)   # :47-48
        o[sl, 0:2] = xy
        o[sl, 2:4] = wh
        row += length
    return o


# --------------------------------------------------------------------------
# yoloDetector.py:104-133  __process_output
# --------------------------------------------------------------------------
def process_output(output, model_type="yolov8", box_score=0.4, input_hw=None):
    """output: the engine's tensor after squeeze(0): (4+nc, A) for v8-family,
    (A, 5+nc) for v5-family, float32.

    Returns (boxes_xyxy f64 [N,4], class_ids int64 [N], confs f64 [N],
             anchor_idx int64 [N]) in anchor order.
    """
    out = np.asarray(output, dtype=np.float32)
    if model_type in V8_FAMILY:
        det = out.T                                   # :115
        probs = det[:, 4:]                            # :122
    else:
        det = out
        if model_type == "yolov5_lite":
            det = lite_postprocess(det, input_hw)         # :117
        probs = det[:, 5:] * det[:, 4:5]              # :124  fp32 product
    if det.shape[0] == 0:
        z = np.zeros((0,), np.int64)
        return np.zeros((0, 4), np.float64), z, np.zeros((0,), np.float64), z
    cls = np.argmax(probs, axis=1)                    # :126 first max
    conf32 = probs[np.arange(det.shape[0]), cls]
    conf = conf32.astype(np.float64)                  # :127 float(...)
    keep = conf > float(box_score)                    # :128 strict
    idx = np.nonzero(keep)[0]
    x = det[idx, 0].astype(np.float64)
    y = det[idx, 1].astype(np.float64)
    w = det[idx, 2].astype(np.float64)
    h = det[idx, 3].astype(np.float64)
    boxes = np.stack([x - 0.5 * w, y - 0.5 * h, x + 0.5 * w, y + 0.5 * h], axis=-1)  # :132
    return boxes, cls[idx].astype(np.int64), conf[idx], idx.astype(np.int64)


# --------------------------------------------------------------------------
# utils.py:70-87  Scaler.convert_boxes_coordinate (in xyxy -> out xywh)
# --------------------------------------------------------------------------
def convert_boxes_coordinate(boxes_xyxy, lb):
    b = np.array(boxes_xyxy, dtype=np.float64).reshape(-1, 4).copy()
    if b.size > 0:
        ratioh, ratiow = lb["ratio"]
        padh, padw = lb["pad"]
        b[:, [0, 2]] = (b[:, [0, 2]] - padw) * ratiow
        b[:, [1, 3]] = (b[:, [1, 3]] - padh) * ratioh
        b[:, 2:4] = b[:, 2:4] - b[:, 0:2]
    return b


# --------------------------------------------------------------------------
# utils.py:161-256  NMS.fast_soft_nms  -- production; degenerates to hard NMS
# with the "+1" area convention and the lossy view-"swap" (SURVEY finding 1)
# --------------------------------------------------------------------------
def fast_soft_nms(dets_xywh, scores, iou_thr=0.45, score_thr=0.001, dets_type="xywh"):
    d = np.array(dets_xywh, dtype=np.float64).reshape(-1, 4).copy()
    sc = np.array(scores, dtype=np.float64).reshape(-1).copy()
    N = d.shape[0]
    if N == 0:
        return np.zeros((0,), np.int32)                       # :190-191 ([] there)
    if dets_type == "xywh":
        d[:, 2:4] = d[:, 0:2] + d[:, 2:4]                     # :187
    if N == 1:
        return np.zeros(1, np.int32)                          # :197-198
    idx = np.arange(N, dtype=np.float64)                      # :202-203 (fp64 col)
    # column labels are swapped in the reference (:206-209); symmetric.
    c0, c1, c2, c3 = d[:, 0].copy(), d[:, 1].copy(), d[:, 2].copy(), d[:, 3].copy()
    areas = (c3 - c1 + 1) * (c2 - c0 + 1)                     # :211
    for i in range(N):
        pos = i + 1
        tscore = sc[i]
        tarea = areas[i]
        if i != N - 1:
            maxpos = pos + int(np.argmax(sc[pos:]))           # first max
            maxscore = sc[maxpos]
        else:
            maxscore = sc[-1]
            maxpos = 0
        if tscore < maxscore:                                 # :225 strict
            # :226 -- tBD is a *view* of row i, so row i <- row maxpos and
            # row maxpos is re-assigned its own (already copied) values.
            c0[i], c1[i], c2[i], c3[i], idx[i] = c0[maxpos], c1[maxpos], c2[maxpos], c3[maxpos], idx[maxpos]
            sc[i], sc[maxpos] = sc[maxpos], tscore            # :227 true swap
            areas[i], areas[maxpos] = areas[maxpos], tarea    # :228 true swap
        if pos < N:
            xx1 = np.maximum(c1[i], c1[pos:])
            yy1 = np.maximum(c0[i], c0[pos:])
            xx2 = np.minimum(c3[i], c3[pos:])
            yy2 = np.minimum(c2[i], c2[pos:])
            w = np.maximum(0.0, xx2 - xx1 + 1)
            h = np.maximum(0.0, yy2 - yy1 + 1)
            inter = w * h
            ovr = inter / (areas[i] + areas[pos:] - inter)
            weight = np.ones_like(ovr)
            weight[ovr > iou_thr] = 0                         # :247-249 (method str != int)
            sc[pos:] = weight * sc[pos:]
    return idx[sc > score_thr].astype(np.int32)               # :254-256


# --------------------------------------------------------------------------
# utils.py:105-159  NMS.fast_nms -- alternative (commented call yoloDetector.py:138)
# Tie order of scores.argsort()[::-1] is unspecified in the reference
# (unstable sort); this restatement fixes it to reversed *stable* ascending
# argsort, i.e. among equal scores the higher original index comes first.
# --------------------------------------------------------------------------
def fast_nms(dets_xywh, scores, iou_thr=0.45, dets_type="xywh"):
    d = np.array(dets_xywh, dtype=np.float64).reshape(-1, 4).copy()
    sc = np.array(scores, dtype=np.float64).reshape(-1)
    N = d.shape[0]
    if N == 0:
        return np.zeros((0,), np.int64)
    if dets_type == "xywh":
        d[:, 2:4] = d[:, 0:2] + d[:, 2:4]
    if N == 1:
        return np.zeros(1, np.int64)
    x1, y1, x2, y2 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    areas = (x2 - x1) * (y2 - y1)                             # :139 no +1
    order = np.argsort(sc, kind="stable")[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        r = order[1:]
        xx1 = np.maximum(x1[i], x1[r]); yy1 = np.maximum(y1[i], y1[r])
        xx2 = np.minimum(x2[i], x2[r]); yy2 = np.minimum(y2[i], y2[r])
        w = np.maximum(0.0, xx2 - xx1); h = np.maximum(0.0, yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[r] - inter)
        order = r[ovr <= iou_thr]                             # :156-157
    return np.asarray(keep, dtype=np.int64)


# --------------------------------------------------------------------------
# yoloDetector.py:135-157 get_nms_results + core.py:18-23 RectInfo.tolist
# --------------------------------------------------------------------------
def rect_infos(boxes_xywh, confs, class_ids, keep):
    """Gather survivors.  Returns dict of arrays in keep order:
    xywh f64 [K,4], conf f64 [K], class_id int64 [K], xyxy_int int64 [K,4]."""
    keep = np.asarray(keep, dtype=np.int64)
    b = np.asarray(boxes_xywh, np.float64).reshape(-1, 4)[keep] if keep.size else np.zeros((0, 4))
    conf = np.asarray(confs, np.float64)[keep] if keep.size else np.zeros((0,))
    cid = np.asarray(class_ids, np.int64)[keep] if keep.size else np.zeros((0,), np.int64)
    xyxy = np.stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]], axis=-1) if keep.size else np.zeros((0, 4))
    xyxy_int = np.trunc(xyxy).astype(np.int64)                # int() truncates toward zero
    return dict(xywh=b, conf=conf, class_id=cid, xyxy_int=xyxy_int)


def detect_post(output, lb, model_type="yolov8", box_score=0.4, iou_thr=0.45, nms_mode="reference", input_hw=None):
    """Full chain of YoloDetector.DetectFrame after the engine (yoloDetector.py:164-168)."""
    boxes, cls, conf, aidx = process_output(output, model_type, box_score, input_hw)
    xywh = convert_boxes_coordinate(boxes, lb)
    if nms_mode == "reference":
        keep = fast_soft_nms(xywh, conf, iou_thr)
    else:
        keep = fast_nms(xywh, conf, iou_thr)
    r = rect_infos(xywh, conf, cls, keep)
    r.update(keep=np.asarray(keep, np.int64), cand_xywh=xywh, cand_conf=conf, cand_cls=cls, cand_anchor=aidx)
    return r
