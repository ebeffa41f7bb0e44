"""Oracle: UFLDv2 row/column-anchor lane decoder + ego-lane area.

TEST INFRASTRUCTURE (see oracle/__init__.py).  NumPy restatement of
  TrafficLaneDetector/ufldDetector/ultrafastLaneDetectorV2.py:15-19   (_softmax)
  .../ultrafastLaneDetectorV2.py:21-55                                (ModelConfig)
  .../ultrafastLaneDetectorV2.py:114-181                              (__process_output)
  TrafficLaneDetector/ufldDetector/core.py:102-158                    (status / area / polyfit adjust)
  .../ultrafastLaneDetector.py:16-40, 96-139                          (UFLD v1: ModelConfig, __process_output)
"""
import numpy as np


class ModelConfig:
    """ultrafastLaneDetectorV2.py:21-55"""

    def __init__(self, name="culane"):
        if name == "tusimple":
            self.img_w, self.img_h, self.griding_num, self.crop_ratio = 800, 320, 100, 0.8
            self.row_anchor = np.linspace(160, 710, 56) / 720
            self.col_anchor = np.linspace(0, 1, 41)
        elif name == "culane":
            self.img_w, self.img_h, self.griding_num, self.crop_ratio = 1600, 320, 200, 0.6
            self.row_anchor = np.linspace(0.42, 1, 72)
            self.col_anchor = np.linspace(0, 1, 81)
        elif name == "curvelanes":                          # :41-47
            self.img_w, self.img_h, self.griding_num, self.crop_ratio = 1600, 800, 200, 0.8
            self.row_anchor = np.linspace(0.4, 1, 72)
            self.col_anchor = np.linspace(0, 1, 81)
        else:
            raise ValueError(name)
        self.num_lanes = 4


def _softmax(x):
    x = x - np.max(x, axis=-1, keepdims=True)
    e = np.exp(x)
    return e / np.sum(e, axis=-1, keepdims=True)


def process_output(outputs, cfg, img_w, img_h, local_width=1):
    """outputs = [loc_row (1,G_r,R,4), loc_col (1,G_c,C,4), exist_row (1,2,R,4), exist_col (1,2,C,4)] fp32.

    Returns (lanes_points: list of 4 lists of (x:int, y:int), lanes_detected: list of 4 bool),
    order left-side, left-ego, right-ego, right-side (:143-145,181).
    """
    loc_row, loc_col, exist_row, exist_col = [np.asarray(o, np.float32) for o in outputs]
    _, G_r, R, _ = loc_row.shape
    _, G_c, C, _ = loc_col.shape
    max_row = loc_row.argmax(1)
    valid_row = exist_row.argmax(1)
    max_col = loc_col.argmax(1)
    valid_col = exist_col.argmax(1)
    pts = {"left-side": [], "left-ego": [], "right-ego": [], "right-side": []}
    det = {"left-side": False, "left-ego": False, "right-ego": False, "right-side": False}
    for i in (1, 2):
        tmp = []
        if valid_row[0, :, i].sum() > R / 2:                                   # :148
            for k in range(R):
                if valid_row[0, k, i]:
                    m = int(max_row[0, k, i])
                    ind = list(range(max(0, m - local_width), min(G_r - 1, m + local_width) + 1))
                    o = (_softmax(loc_row[0, ind, k, i]) * np.array(ind, dtype=np.float64)).sum() + 0.5
                    o = o / (G_r - 1) * img_w
                    tmp.append((int(o), int(cfg.row_anchor[k] * img_h)))
        name = "left-ego" if i == 1 else "right-ego"
        pts[name].extend(tmp)
        det[name] = len(tmp) > 2
    for i in (0, 3):
        tmp = []
        if valid_col[0, :, i].sum() > C / 4:                                   # :166
            for k in range(C):
                if valid_col[0, k, i]:
                    m = int(max_col[0, k, i])
                    ind = list(range(max(0, m - local_width), min(G_c - 1, m + local_width) + 1))
                    o = (_softmax(loc_col[0, ind, k, i]) * np.array(ind, dtype=np.float64)).sum() + 0.5
                    o = o / (G_c - 1) * img_h
                    tmp.append((int(cfg.col_anchor[k] * img_w), int(o)))
        name = "left-side" if i == 0 else "right-side"
        pts[name].extend(tmp)
        det[name] = len(tmp) > 2
    return list(pts.values()), list(det.values())


# core.py:102-141
def adjust_lanes_points(left, right, image_height):
    if len(left) > 1 and len(left[1]) != 0:
        leftx, lefty = list(zip(*left))
        if len(lefty) > 10:
            left_fit = np.polyfit(lefty, leftx, 2)
        else:
            return left, right
    else:
        return left, right
    if len(right) != 0:
        rightx, righty = list(zip(*right))
        if len(righty) > 10:
            right_fit = np.polyfit(righty, rightx, 2)
        else:
            return left, right
    else:
        return left, right
    maxy = image_height - 1
    miny = image_height // 3
    if len(lefty):
        maxy = max(maxy, np.max(lefty)); miny = min(miny, np.min(lefty))
    if len(righty):
        maxy = max(maxy, np.max(righty)); miny = min(miny, np.min(righty))
    fity = np.linspace(miny, maxy, image_height)
    lfx = left_fit[0] * fity ** 2 + left_fit[1] * fity + left_fit[2]
    rfx = right_fit[0] * fity ** 2 + right_fit[1] * fity + right_fit[2]
    fl = [(int(l), int(y)) for l, y in zip(lfx, fity) if (y >= min(lefty) and l >= 0)]
    fr = [(int(r), int(y)) for r, y in zip(rfx, fity) if (y >= min(righty) and r >= 0)]
    return fl, fr


# core.py:143-158
def lanes_area(lanes_points, lanes_status, img_height, adjust=True):
    """Returns (area_status: bool, area_points: (P,2) int array or empty)."""
    status = False
    if lanes_status != [] and len(lanes_status) % 2 == 0:
        index = len(lanes_status) // 2
        if lanes_status[index - 1] and lanes_status[index]:
            status = True
    area = np.zeros((0, 2), np.int64)
    if status:
        index = len(lanes_points) // 2
        if adjust:
            l, r = adjust_lanes_points(lanes_points[index - 1], lanes_points[index], img_height)
        else:
            l, r = lanes_points[index - 1], lanes_points[index]
        area = np.vstack((l, np.flipud(r)))
    return status, area


# ------------------------------------------------------------------------------------------------
# UFLD (v1)
# ------------------------------------------------------------------------------------------------
class ModelConfigV1:
    """ultrafastLaneDetector.py:16-40"""

    def __init__(self, name="tusimple"):
        if name == "tusimple":
            self.img_w, self.img_h, self.griding_num, self.cls_num_per_lane = 1280, 720, 100, 56
            self.row_anchor = np.linspace(64, 284, 56)
        elif name == "culane":
            self.img_w, self.img_h, self.griding_num, self.cls_num_per_lane = 1640, 590, 200, 18
            self.row_anchor = [round(v) for v in np.linspace(121, 287, 18)]
        else:
            raise ValueError(name)
        self.num_lanes = 4


def process_output_v1(output, cfg, input_w, input_h, src_w, src_h):
    """output: the engine's single tensor (1, G+1, K, L) fp32.  Returns (lanes: L lists of (x, y) ints, detected: L bools).

    :101-109  flip the anchor axis; softmax over the G real cells in fp32 (scipy.special.softmax = exp(x - max) / sum);
              location = sum(prob * (1..G)) in fp64; rows whose argmax over all G+1 cells is the last ("no lane") -> 0
    :113-139  a lane is detected when more than 2 anchors are non-zero; points in flipped-anchor order."""
    G, K = cfg.griding_num, cfg.cls_num_per_lane
    raw = np.asarray(output, np.float32).reshape(G + 1, K, -1)[:, ::-1, :]
    z = raw[:G]
    e = np.exp(z - z.max(axis=0, keepdims=True))
    prob = e / e.sum(axis=0, keepdims=True)
    loc = (prob * np.arange(1, G + 1).reshape(-1, 1, 1)).sum(axis=0)
    loc[raw.argmax(axis=0) == G] = 0
    step = np.linspace(0, input_w - 1, G)
    col_w = step[1] - step[0]
    w_ratio, h_ratio = src_w / cfg.img_w, src_h / cfg.img_h              # :80
    lanes, detected = [], []
    for lane in range(loc.shape[1]):
        pts = []
        ok = int((loc[:, lane] != 0).sum()) > 2
        if ok:
            for k in range(K):
                v = loc[k, lane]
                if v > 0:
                    x = v * col_w * cfg.img_w / input_w - 1
                    y = cfg.img_h * (cfg.row_anchor[K - 1 - k] / input_h) - 1
                    pts.append((int(x * w_ratio), int(y * h_ratio)))
        lanes.append(pts)
        detected.append(ok)
    return lanes, detected
