"""Seeded synthetic inputs shared by make_golden.py (build container, needs the
reference) and the parity tests (anywhere).  No reference dependency.  Fixtures
store sha1 digests of these inputs so a drifting RNG stream is detected."""
import hashlib
import numpy as np


def digest(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def synth_v8_head(seed, n_hot=60, n_dup=20, A=8400, nc=80):
    """SURVEY 8c recipe, parameterised."""
    rng = np.random.default_rng(seed)
    out = np.zeros((4 + nc, A), np.float32)
    out[0] = rng.uniform(50, 590, A); out[1] = rng.uniform(180, 460, A)
    out[2] = rng.uniform(20, 120, A); out[3] = rng.uniform(20, 100, A)
    out[4:] = rng.uniform(0, 0.05, (nc, A))
    hot = rng.choice(A, n_hot, replace=False)
    for a in hot:
        out[4 + rng.integers(nc), a] = rng.uniform(0.35, 0.95)
    nd = min(n_dup, n_hot // 3)
    out[0:4, hot[nd:2 * nd]] = out[0:4, hot[:nd]] + rng.normal(0, 2, (4, nd)).astype(np.float32)
    return out


def synth_v5_head(seed, n_hot=50, A=25200, nc=80):
    rng = np.random.default_rng(seed)
    out = np.zeros((A, 5 + nc), np.float32)
    out[:, 0] = rng.uniform(50, 590, A); out[:, 1] = rng.uniform(50, 590, A)
    out[:, 2] = rng.uniform(20, 120, A); out[:, 3] = rng.uniform(20, 100, A)
    q = 2048.0  # dyadic rationals -> fp32 product exact
    out[:, 4] = np.round(rng.uniform(0, 0.3, A) * q) / q
    out[:, 5:] = np.round(rng.uniform(0, 0.2, (A, nc)) * q) / q
    hot = rng.choice(A, n_hot, replace=False)
    for a in hot:
        out[a, 4] = np.round(rng.uniform(0.7, 1.0) * q) / q
        out[a, 5 + rng.integers(nc)] = np.round(rng.uniform(0.6, 1.0) * q) / q
    nd = n_hot // 3
    out[hot[nd:2 * nd], 0:4] = out[hot[:nd], 0:4] + rng.normal(0, 2, (nd, 4)).astype(np.float32)
    return out


def synth_ufld(seed, lanes=((1, 60, .4), (2, 140, -.4)), cols=(), boost=True):
    rng = np.random.default_rng(seed)
    loc_row = rng.normal(0, 1, (1, 200, 72, 4)).astype(np.float32)
    loc_col = rng.normal(0, 1, (1, 100, 81, 4)).astype(np.float32)
    exist_row = rng.normal(0, 1, (1, 2, 72, 4)).astype(np.float32)
    exist_col = rng.normal(0, 1, (1, 2, 81, 4)).astype(np.float32)
    for lane, x0, slope in lanes:
        for k in range(72):
            g = int(x0 + slope * k)
            g = max(0, min(199, g))
            loc_row[0, g, k, lane] += 8
            if g + 1 <= 199:
                loc_row[0, g + 1, k, lane] += 6
            exist_row[0, 1, k, lane] += 5
    for lane, y0, slope in cols:
        for k in range(81):
            g = int(y0 + slope * k)
            g = max(0, min(99, g))
            loc_col[0, g, k, lane] += 8
            if g >= 1:
                loc_col[0, g - 1, k, lane] += 5
            exist_col[0, 1, k, lane] += 5
    return [loc_row, loc_col, exist_row, exist_col]


def synth_ufld_curve(seed, lanes=((1, 60, .4), (2, 140, -.4)), cols=((0, 30, .5),), extra=((5, 100, .2), (9, 20, .1))):
    """CurveLanes-configuration heads (configs/curvelanes_res18.py): (1,200,72,10), (1,100,41,10), (1,2,72,10), (1,2,41,10).  `extra`
    lanes (index >= 4) carry strong responses too: the reference ignores them (it decodes lanes 1,2 / 0,3 only)."""
    rng = np.random.default_rng(seed)
    R, C, NL = 72, 41, 10
    loc_row = rng.normal(0, 1, (1, 200, R, NL)).astype(np.float32)
    loc_col = rng.normal(0, 1, (1, 100, C, NL)).astype(np.float32)
    exist_row = rng.normal(0, 1, (1, 2, R, NL)).astype(np.float32)
    exist_col = rng.normal(0, 1, (1, 2, C, NL)).astype(np.float32)
    for lane, x0, slope in tuple(lanes) + tuple(extra):
        for k in range(R):
            g = max(0, min(199, int(x0 + slope * k)))
            loc_row[0, g, k, lane] += 8
            if g + 1 <= 199:
                loc_row[0, g + 1, k, lane] += 6
            exist_row[0, 1, k, lane] += 5
    for lane, y0, slope in cols:
        for k in range(C):
            g = max(0, min(99, int(y0 + slope * k)))
            loc_col[0, g, k, lane] += 8
            if g >= 1:
                loc_col[0, g - 1, k, lane] += 5
            exist_col[0, 1, k, lane] += 5
    return [loc_row, loc_col, exist_row, exist_col]


def curve_cases():
    """(tag, heads, img_w, img_h)"""
    return [("c1", synth_ufld_curve(40), 1280, 720),
            ("c2", synth_ufld_curve(41, lanes=((1, 20, 1.2), (2, 190, -1.5)), cols=((0, 30, .5), (3, 80, -.6))), 1920, 1080),
            ("c3", synth_ufld_curve(42, lanes=(), cols=()), 1280, 720),
            ("c4", synth_ufld_curve(43, lanes=((2, 100, 0.0),), cols=((3, 99, 0.0),), extra=((4, 50, .3), (7, 150, -.3))), 2560, 1440)]


def effdet_cases():
    """(tag, boxes (n,4) f32 xyxy in input pixels, ids (n,) i32, confs (n,) f32, source (h, w), input (h, w), box_score)"""
    out = []
    for tag, seed, n, src, inp, thr in (("e1", 50, 40, (720, 1280), (512, 512), 0.6), ("e2", 51, 100, (1080, 1920), (640, 640), 0.3),
                                        ("e3", 52, 0, (720, 1280), (512, 512), 0.6), ("e4", 53, 17, (900, 600), (768, 768), 0.6),
                                        ("e5", 54, 64, (512, 512), (512, 512), 0.5)):
        rng = np.random.default_rng(seed)
        x1 = rng.uniform(0, inp[1] - 40, n); y1 = rng.uniform(0, inp[0] - 40, n)
        boxes = np.stack([x1, y1, x1 + rng.uniform(5, 200, n), y1 + rng.uniform(5, 200, n)], 1).astype(np.float32)
        ids = rng.integers(0, 90, n).astype(np.int32)
        confs = np.sort(rng.uniform(0.05, 0.99, n).astype(np.float32))[::-1].copy()      # the graph's NMS returns them by score
        if n > 4:
            confs[3] = np.float32(thr)            # exactly at the threshold: kept (`conf < box_score` is false)
            confs[4] = np.nextafter(np.float32(thr), np.float32(0))   # one ulp below the float32 image of the threshold
        out.append((tag, boxes, ids, confs, src, inp, thr))
    return out


def track_scene(seed, n_obj, n_frames, drop=0.1, W=1280, H=720):
    """Constant-velocity rectangles + N(0,1) jitter + dropout; int xyxy like RectInfo.tolist()."""
    rng = np.random.default_rng(seed)
    cx = rng.uniform(100, W - 100, n_obj); cy = rng.uniform(100, H - 100, n_obj)
    w = rng.uniform(40, 160, n_obj); h = rng.uniform(40, 140, n_obj)
    vx = rng.uniform(-6, 6, n_obj); vy = rng.uniform(-3, 3, n_obj)
    cls = rng.integers(0, 3, n_obj)
    base_score = rng.uniform(0.42, 0.95, n_obj)
    frames = []
    for f in range(n_frames):
        boxes, scores, ids = [], [], []
        for o in range(n_obj):
            if rng.uniform() < drop:
                continue
            j = rng.normal(0, 1, 4)
            x1 = cx[o] + vx[o] * f - w[o] / 2 + j[0]; y1 = cy[o] + vy[o] * f - h[o] / 2 + j[1]
            x2 = cx[o] + vx[o] * f + w[o] / 2 + j[2]; y2 = cy[o] + vy[o] * f + h[o] / 2 + j[3]
            boxes.append([int(x1), int(y1), int(x2), int(y2)])
            s = float(np.clip(base_score[o] + rng.normal(0, 0.05), 0.401, 0.99))
            scores.append(float(np.float32(s)))
            ids.append(int(cls[o]))
        frames.append(dict(boxes=boxes, scores=scores, ids=ids))
    return frames



LB720 = dict(target=(640, 640), old=(720, 1280), new=(361, 640), pad=(139, 0))
LBSQ = dict(target=(640, 640), old=(640, 640), new=(640, 640), pad=(0, 0))
LBPORT = dict(target=(640, 640), old=(1280, 720), new=(640, 360), pad=(0, 140))


def synth_lite_head(seed, input_hw=(640, 640), n_hot=60, nc=80):
    """Raw YOLOv5-lite head (A, 5+nc): sigmoid-range xy/wh offsets (not yet grid-decoded), dyadic obj/cls."""
    rng = np.random.default_rng(seed)
    A = sum(3 * (input_hw[0] // s) * (input_hw[1] // s) for s in (8, 16, 32))
    out = np.zeros((A, 5 + nc), np.float32)
    out[:, 0:2] = rng.uniform(0.02, 0.98, (A, 2))
    out[:, 2:4] = rng.uniform(0.05, 0.95, (A, 2))
    q = 2048.0
    out[:, 4] = np.round(rng.uniform(0, 0.3, A) * q) / q
    out[:, 5:] = np.round(rng.uniform(0, 0.2, (A, nc)) * q) / q
    hot = rng.choice(A, n_hot, replace=False)
    hot[:6] = [0, A - 1, 3 * (input_hw[0] // 8) * (input_hw[1] // 8) - 1, 3 * (input_hw[0] // 8) * (input_hw[1] // 8),
               (input_hw[0] // 8) * (input_hw[1] // 8), A - 3 * (input_hw[0] // 32) * (input_hw[1] // 32)]   # level / anchor seams
    for a in hot:
        out[a, 4] = np.round(rng.uniform(0.7, 1.0) * q) / q
        out[a, 5 + rng.integers(nc)] = np.round(rng.uniform(0.6, 1.0) * q) / q
    return out


def lite_cases():
    """(tag, head, input_hw, letterbox, box_score, iou); the 384x640 case exercises the reference's (n % h, n // h) grid."""
    lb_wide = dict(target=(384, 640), old=(720, 1280), new=(217, 640), pad=(83, 0))
    return [("lite_sq", synth_lite_head(21), (640, 640), LB720, 0.4, 0.45),
            ("lite_sq2", synth_lite_head(22, (640, 640), 300), (640, 640), LBSQ, 0.4, 0.45),
            ("lite_wide", synth_lite_head(23, (384, 640), 120), (384, 640), lb_wide, 0.4, 0.45)]


def synth_ufld1(seed, G=100, K=56, spans=((0, 56), (10, 40), (20, 22), (0, 0))):
    """UFLD v1 head (1, G+1, K, 4): lane l has a softmax bump drifting across the grid on anchors [spans[l]) and votes
    "no lane" (cell G) elsewhere; the third default lane has only two points (not detected, ultrafastLaneDetector.py:123)."""
    rng = np.random.default_rng(seed)
    out = rng.normal(0, 0.6, (1, G + 1, K, 4)).astype(np.float32)
    g = np.arange(G, dtype=np.float64)
    for l, (a, b) in enumerate(spans):
        for k in range(K):
            if a <= k < b:
                c = (0.15 + 0.2 * l + 0.5 * (k - a) / max(1, b - a)) * G + rng.normal(0, 0.4)
                out[0, :G, k, l] += (9.0 * np.exp(-0.5 * ((g - c) / 1.3) ** 2)).astype(np.float32)
                out[0, G, k, l] -= 2.0
            else:
                out[0, G, k, l] += 7.0
    return out


def ufld1_cases():
    """(tag, config name, head, input (w, h), source (w, h))"""
    return [("u1_tu", "tusimple", synth_ufld1(31), (800, 288), (1280, 720)),
            ("u1_tu_hd", "tusimple", synth_ufld1(32, spans=((3, 50), (0, 56), (0, 3), (30, 56))), (800, 288), (1920, 1080)),
            ("u1_cu", "culane", synth_ufld1(33, 200, 18, ((0, 18), (2, 15), (5, 7), (0, 0))), (800, 288), (1640, 590)),
            ("u1_none", "tusimple", synth_ufld1(34, spans=((0, 0), (0, 0), (0, 2), (0, 1))), (800, 288), (1280, 720))]


def yolo_cases():
    """(tag, model_type, head, letterbox, box_score, iou)"""
    return [("v8_s1", "YOLOV8", synth_v8_head(1), LB720, 0.4, 0.45),
            ("v8_s2", "YOLOV8", synth_v8_head(2, 200, 60), LB720, 0.4, 0.45),
            ("v8_s3", "YOLOV8", synth_v8_head(3, 12, 3), LBSQ, 0.25, 0.5),
            ("v8_s4", "YOLOV8", synth_v8_head(4, 600, 200), LBPORT, 0.4, 0.45),
            ("v8_none", "YOLOV8", synth_v8_head(5, 0, 0), LB720, 0.4, 0.45),
            ("v5_s1", "YOLOV5", synth_v5_head(11), LBSQ, 0.4, 0.45),
            ("v5_s2", "YOLOV5", synth_v5_head(12, 150), LB720, 0.4, 0.45)]


def ufld_cases():
    """(tag, [loc_row, loc_col, exist_row, exist_col], img_w, img_h)"""
    return [("l1", synth_ufld(0), 1280, 720),
            ("l2", synth_ufld(1, lanes=((1, 20, 1.2), (2, 190, -1.5)), cols=((0, 30, .5), (3, 80, -.6))), 1280, 720),
            ("l3", synth_ufld(2, lanes=()), 1920, 1080),
            ("l4", synth_ufld(3, lanes=((1, 0, 0.0), (2, 199, 0.0)), cols=((0, 0, 0.0), (3, 99, 0.0))), 1640, 590),
            ("l5", synth_ufld(4, lanes=((1, 70, .3),)), 1280, 720)]
