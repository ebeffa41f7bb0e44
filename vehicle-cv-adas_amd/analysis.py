"""The consumers of the hot path's outputs (SURVEY.md 8f rows f2 + f3), Linux-runnable and without cv2:

    SingleCamDistanceMeasure   <- ObjectDetector/distanceMeasure.py:7-93   (distance from box height, nearest object inside the ego lane)
    PerspectiveTransformation  <- ufldDetector/perspectiveTransformation.py:10-214 (homography of lane POINTS, curvature radius, lateral offset)
    TaskConditions             <- taskConditions.py:88-312 (median-window FCWS / LDWS / LKAS state machine; the reference file imports
                                  ctypes.windll at module scope and runs on Windows only)

They take what `detectors.py` produces (RectInfo list, LaneInfo) -- a few hundred bytes per frame -- and, like the reference,
run on the host: after GPU-resident decode/NMS/tracking this is per-frame scalar bookkeeping (<0.2 ms), not a kernel.
Same public names and call shapes as the reference so demo.py:284-296 reads unchanged; image warping and every Draw*
method are out of scope (cv2).

Third-party arithmetic restated here (parity UNPINNED against real OpenCV, cv2 is absent): `perspective_matrix`
(cv2.getPerspectiveTransform: the 8x8 linear system of the four point pairs, solved in float64) and `point_in_polygon`
(cv2.pointPolygonTest(measureDist=False): +1 inside, 0 on an edge/vertex, -1 outside).  Everything else is pinned by
tests/golden/analysis.json.gz, produced by the reference's own classes under stubs (tests/golden/make_golden_analysis.py).
"""
from enum import Enum
from typing import List, Optional

import numpy as np


class CollisionType(Enum):        # ObjectDetector/utils.py:8-12
    UNKNOWN = "Determined ..."
    NORMAL = "Normal Risk"
    PROMPT = "Prompt Risk"
    WARNING = "Warning Risk"


class OffsetType(Enum):           # ufldDetector/utils.py:10-14
    UNKNOWN = "To Be Determined ..."
    RIGHT = "Please Keep Right"
    LEFT = "Please Keep Left"
    CENTER = "Good Lane Keeping"


class CurvatureType(Enum):        # ufldDetector/utils.py:16-22
    UNKNOWN = "To Be Determined ..."
    STRAIGHT = "Keep Straight Ahead"
    EASY_LEFT = "Gentle Left Curve Ahead"
    HARD_LEFT = "Hard Left Curve Ahead"
    EASY_RIGHT = "Gentle Right Curve Ahead"
    HARD_RIGHT = "Hard Right Curve Ahead"


# =====================================================================================
# f3a: distance from a single camera
# =====================================================================================
def point_in_polygon(poly, pt) -> int:
    """cv2.pointPolygonTest(poly, pt, False): +1 inside, 0 on the boundary, -1 outside (even-odd crossing rule)."""
    p = np.asarray(poly, np.float64).reshape(-1, 2)
    n = len(p)
    if n == 0:
        return -1
    x, y = float(pt[0]), float(pt[1])
    inside = False
    for i in range(n):
        x0, y0 = p[i - 1]
        x1, y1 = p[i]
        # on the segment?
        cross = (x1 - x0) * (y - y0) - (y1 - y0) * (x - x0)
        if cross == 0 and min(x0, x1) <= x <= max(x0, x1) and min(y0, y1) <= y <= max(y0, y1):
            return 0
        if (y0 <= y < y1) or (y1 <= y < y0):
            t = (y - y0) / (y1 - y0)
            if x0 + t * (x1 - x0) > x:
                inside = not inside
    return 1 if inside else -1


class SingleCamDistanceMeasure:
    INCH = 0.39                     # 1 cm = 0.39 inch (distanceMeasure.py:9)
    RefSizeDict = {                 # real-world (height, width) in inches (:10-17)
        "person": (160 * INCH, 50 * INCH), "bicycle": (98 * INCH, 65 * INCH), "motorbike": (100 * INCH, 100 * INCH),
        "car": (150 * INCH, 180 * INCH), "bus": (319 * INCH, 250 * INCH), "truck": (346 * INCH, 250 * INCH),
    }

    def __init__(self, object_list=("person", "bicycle", "car", "motorbike", "bus", "truck")):
        self.object_list = list(object_list)
        self.f = 100                # focal length (:21)
        self.distance_points: List[list] = []

    def updateDistance(self, boxes) -> None:
        """[x_centre, y_bottom, metres] for every known object whose box bottom is at or above row 650 (:50-74)."""
        pts = []
        for box in boxes or []:
            xmin, ymin, xmax, ymax = box.tolist()
            ref = self.RefSizeDict.get(box.label) if box.label in self.object_list else None
            if ref is None or ymax > 650 or ymax == ymin:
                continue
            inches = ref[0] * self.f / (ymax - ymin)
            pts.append([(xmax + xmin) // 2, ymax, inches / 12 * 0.3048])
        self.distance_points = pts

    def calcCollisionPoint(self, poly) -> Optional[list]:
        """Nearest measured object whose foot point lies in (or on) the ego-lane polygon (:76-93)."""
        if not self.distance_points or poly is None or len(poly) == 0:
            return None
        for x, y, d in sorted(self.distance_points, key=lambda q: q[2]):
            if point_in_polygon(poly, (x, y)) >= 0:
                return [x, y, d]
        return None


# =====================================================================================
# f2: bird-view geometry of the lane points
# =====================================================================================
def perspective_matrix(src, dst) -> np.ndarray:
    """cv2.getPerspectiveTransform: H (3x3, H[2,2] = 1) with dst ~ H @ src for four point pairs."""
    s = np.asarray(src, np.float64).reshape(4, 2)
    d = np.asarray(dst, np.float64).reshape(4, 2)
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = s[i]
        u, v = d[i]
        A[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    h = np.linalg.solve(A, b)
    return np.append(h, 1.0).reshape(3, 3)


class PerspectiveTransformation:
    def __init__(self, img_size=(1280, 720), logger=None):
        self.img_size = img_size
        self.logger = logger
        w, h = img_size
        self.src = np.float32([(w * 0.3, h * 0.7), (w * 0.2, h), (w * 0.95, h), (w * 0.8, h * 0.7)])      # :24-27 tl, bl, br, tr
        ox = w / 4
        self.dst = np.float32([(ox, 0), (ox, h), (w - ox, h), (w - ox, 0)])                              # :29-34
        self._refresh()

    def _refresh(self):
        self.M = perspective_matrix(self.src, self.dst)
        self.M_inv = perspective_matrix(self.dst, self.src)

    def updateTransformParams(self, left_lanes, right_lanes, type: str = "Default") -> None:
        """Re-anchor the frontal-view trapezoid on the two ego lanes (:39-86)."""
        L = np.asarray(left_lanes if isinstance(left_lanes, list) else np.asarray(left_lanes).tolist(), np.float64).reshape(-1, 2)
        R = np.asarray(right_lanes if isinstance(right_lanes, list) else np.asarray(right_lanes).tolist(), np.float64).reshape(-1, 2)
        if len(L) == 0 or len(R) == 0 or type not in ("Top", "Bottom", "Default"):
            return
        tl, bl, br, tr = (tuple(p) for p in self.src)
        top_y = min(L[:, 1].min(), R[:, 1].min())
        if type in ("Top", "Default"):
            tl, tr = (L[:, 0].max() - 20, top_y), (R[:, 0].min() + 20, top_y)
        if type == "Top":
            bl, br = (bl[0] - 10, bl[1]), (br[0] + 10, br[1])
        elif type == "Bottom":
            bl, br = (L[:, 0].min() - 20, bl[1]), (R[:, 0].max() + 20, br[1])
        else:
            bl, br = (L[:, 0].min() - 5, bl[1]), (R[:, 0].max() + 5, br[1])
        self.src = np.float32([tl, bl, br, tr])
        self._refresh()

    def transformToBirdViewPoints(self, points):
        """Homogeneous transform of (x, y) points, integer-truncated (:120-142)."""
        if points is None or len(points) == 0:
            return []
        p = np.array([[x, y] for x, y in points])
        q = np.einsum('kl, ...l->...k', self.M, np.concatenate([p, np.broadcast_to(1, (*p.shape[:-1], 1))], axis=-1))
        return np.asarray(q[..., :2] / q[..., 2][..., None], dtype='int')

    def calcCurveAndOffset(self, img, left_lanes, right_lanes):
        """((direction 'L'|'R'|'F', curvature radius in m), offset from the lane centre in m)  (:145-214).
        `img` may be the bird-view image or just its (H, W[, C]) shape."""
        shape = img if isinstance(img, (tuple, list)) else img.shape
        if left_lanes is None or right_lanes is None or len(left_lanes) == 0 or len(right_lanes) == 0:
            return (None, None), None
        L = np.squeeze(np.asarray(left_lanes))
        R = np.squeeze(np.asarray(right_lanes))
        lf = np.polyfit(L[:, 1], L[:, 0], 2)
        rf = np.polyfit(R[:, 1], R[:, 0], 2)
        bend = lf[0] if abs(lf[0]) > abs(rf[0]) else rf[0]
        if bend < -0.00015 and L[0, 0] <= L[int(len(L) / 2), 0]:
            direction = "L"
        elif bend > 0.00015 and R[0, 0] >= R[int(len(R) / 2), 0]:
            direction = "R"
        else:
            direction = "F"
        ploty = np.linspace(0, shape[0] - 1, shape[0])
        leftx = lf[0] * ploty ** 2 + lf[1] * ploty + lf[2]
        rightx = rf[0] * ploty ** 2 + rf[1] * ploty + rf[2]
        ym, xm = 30 / 720, 3.7 / 700                               # metres per pixel (:183-184)
        y_eval = np.max(ploty)
        lcr = np.polyfit(ploty * ym, leftx * xm, 2)
        rcr = np.polyfit(ploty * ym, rightx * xm, 2)
        rad = lambda c: ((1 + (2 * c[0] * y_eval * ym + c[1]) ** 2) ** 1.5) / np.absolute(2 * c[0])
        curvature = (rad(lcr) + rad(rcr)) / 2
        lane_width = np.absolute(leftx[719] - rightx[719])        # row 719, as the reference hard-codes (:196-199)
        veh_pos = (leftx[719] + rightx[719]) / 2.
        offset = (veh_pos - shape[1] / 2.) * (3.7 / lane_width)
        return (direction, curvature), offset


# =====================================================================================
# f3b: warning state machine
# =====================================================================================
class _Window(list):
    """Fixed-length FIFO (taskConditions.py:13-36 LimitedList)."""

    def __init__(self, maxlen):
        super().__init__()
        self._maxlen = maxlen

    def full(self):
        return len(self) >= self._maxlen

    def append(self, element):
        if len(self) == self._maxlen:
            del self[0]
        super().append(element)


class TaskConditions:
    def __init__(self):
        self.collision_msg = CollisionType.UNKNOWN
        self.offset_msg = OffsetType.UNKNOWN
        self.curvature_msg = CurvatureType.UNKNOWN
        self.vehicle_collision_record = _Window(5)
        self.vehicle_offset_record = _Window(5)
        self.vehicle_curvature_record = _Window(10)
        self.transform_status = None
        self.toggle_status = "Default"
        self.toggle_oscillator_status = [False, False]
        self.toggle_status_counter = {"Offset": 0, "Curvae": 0, "BirdViewAngle": 0}

    # ---- helpers (:101-178)
    def _calibration_curve(self, vehicle_curvature, frequency=3, curvae_thres=15000):
        c = self.toggle_status_counter
        if c["BirdViewAngle"] > frequency:
            c["BirdViewAngle"] = 0
            self.toggle_status = "Default"
        else:
            c["BirdViewAngle"] = c["BirdViewAngle"] + 1 if vehicle_curvature >= curvae_thres else 0

    def _calc_deviation(self, offset, offset_thres):
        if abs(offset) <= offset_thres:
            return OffsetType.CENTER
        if offset > 0 and self.curvature_msg not in {CurvatureType.HARD_LEFT, CurvatureType.EASY_LEFT}:
            return OffsetType.RIGHT
        if offset < 0 and self.curvature_msg not in {CurvatureType.HARD_RIGHT, CurvatureType.EASY_RIGHT}:
            return OffsetType.LEFT
        return OffsetType.UNKNOWN

    def _calc_direction(self, curvature, curvae_dir, curvae_thres):
        if curvature <= curvae_thres:
            if curvae_dir == "L" and self.curvature_msg != CurvatureType.EASY_RIGHT:
                return CurvatureType.HARD_LEFT
            if curvae_dir == "R" and self.curvature_msg != CurvatureType.EASY_LEFT:
                return CurvatureType.HARD_RIGHT
            return CurvatureType.UNKNOWN
        return {"L": CurvatureType.EASY_LEFT, "R": CurvatureType.EASY_RIGHT}.get(curvae_dir, CurvatureType.STRAIGHT)

    # ---- public surface (:180-312)
    def CheckStatus(self) -> bool:
        if self.curvature_msg == CurvatureType.UNKNOWN and self.offset_msg == OffsetType.UNKNOWN:
            self.toggle_oscillator_status = [False, False]
        if self.toggle_status != self.transform_status:
            self.transform_status = self.toggle_status
            self.toggle_status = None
            return True
        return False

    def UpdateOffsetStatus(self, vehicle_offset, offset_thres=0.65) -> None:
        rec, cnt = self.vehicle_offset_record, self.toggle_status_counter
        if vehicle_offset is None:
            self.offset_msg = OffsetType.UNKNOWN
            rec.clear()
            return
        rec.append(vehicle_offset)
        if not rec.full():
            self.offset_msg = OffsetType.UNKNOWN
            return
        self.offset_msg = self._calc_deviation(np.median(rec), offset_thres)
        if cnt["Offset"] < 10:
            cnt["Offset"] += 1
            return
        if all(v > 0.2 for v in rec):
            self.toggle_oscillator_status[0] = True
            cnt["Offset"] = 0
        if all(v < -0.2 for v in rec):
            self.toggle_oscillator_status[1] = True
            cnt["Offset"] = 0
        if all(self.toggle_oscillator_status):
            self.toggle_status = "Top"
            self.toggle_oscillator_status = [False, False]
        else:
            cnt["Offset"] = 0

    def UpdateRouteStatus(self, vehicle_direction, vehicle_curvature, curvae_thres=500) -> None:
        rec, cnt = self.vehicle_curvature_record, self.toggle_status_counter
        if vehicle_curvature is None:
            rec.clear()
            self.curvature_msg = CurvatureType.UNKNOWN
            return
        if vehicle_direction is not None and self.offset_msg == OffsetType.CENTER:
            rec.append([vehicle_direction, vehicle_curvature])
            if rec.full():
                # Bug-compatible with taskConditions.py:262: `key=record.count` looks a direction STRING up in a list of
                # [direction, curvature] pairs, finds none, so every key is 0 and max() returns the first element of the set's
                # iteration order -- hash-seed dependent whenever the window mixes directions (tests pin PYTHONHASHSEED=0).
                avg_direction = max(set(np.squeeze(rec)[:, 0]), key=rec.count)
                avg_curvature = np.median([int(float(r[1])) for r in rec])
                self.curvature_msg = self._calc_direction(avg_curvature, avg_direction, curvae_thres)
                if cnt["Curvae"] >= 10:
                    if (self.curvature_msg != CurvatureType.STRAIGHT and abs(self.vehicle_offset_record[-1]) < 0.2
                            and not any(self.toggle_oscillator_status)):
                        self.toggle_status = "Bottom"
                    else:
                        cnt["Curvae"] = 0
                else:
                    cnt["Curvae"] += 1
            else:
                self.curvature_msg = CurvatureType.UNKNOWN
        else:
            rec.clear()
            self.curvature_msg = CurvatureType.UNKNOWN
        self._calibration_curve(vehicle_curvature)

    def UpdateCollisionStatus(self, vehicle_distance, lane_area, distance_thres=1.5) -> None:
        if vehicle_distance is None:        # nothing in the ego lane: the median window restarts (taskConditions.py:307-312)
            self.collision_msg = CollisionType.NORMAL if lane_area else CollisionType.UNKNOWN
            self.vehicle_collision_record.clear()
            return
        self.vehicle_collision_record.append(vehicle_distance[2])
        if self.vehicle_collision_record.full():
            d = np.median(self.vehicle_collision_record)
            if d <= distance_thres:
                self.collision_msg = CollisionType.WARNING
            elif d <= 2 * distance_thres:
                self.collision_msg = CollisionType.PROMPT
            else:
                self.collision_msg = CollisionType.NORMAL
