"""ctypes wrapper over tests/_build/libemu.so (host emulation of the device post-processing logic).
Test scaffolding: lets the CPU suite check the logic of csrc/post_core.h + track_core.h."""
import ctypes as C, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostemu", "emu.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "libemu.so")
INC = os.path.join(ROOT, "vehicle-cv-adas_amd", "csrc")

BTOUT = np.dtype([("tlwh", "f8", 4), ("score", "f8"), ("track_id", "i4"), ("state", "i4"), ("is_activated", "i4"),
                  ("class_id", "i4"), ("frame_id", "i4"), ("start_frame", "i4"), ("tracklet_len", "i4"), ("traj_len", "i4")])


def build():
    deps = [SRC, os.path.join(INC, "post_core.h"), os.path.join(INC, "track_core.h"), os.path.join(INC, "lane_core.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-I", INC, SRC, "-o", OUT])
    return OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.emu_bt_create.restype = C.c_void_p
        _lib.emu_bt_create.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]
        for f in ("emu_bt_destroy", "emu_bt_reset"):
            getattr(_lib, f).argtypes = [C.c_void_p]
        _lib.emu_bt_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.emu_bt_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.emu_bt_trajectories.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        assert _lib.emu_sizeof_btout() == BTOUT.itemsize
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def yolo_post(head, layout, lb, box_score, iou, nms_mode=0, cap=1024, input_hw=(0, 0)):
    head = np.ascontiguousarray(head, np.float32)
    if layout == 0:
        nc, A = head.shape[0] - 4, head.shape[1]
    else:
        A, nc = head.shape[0], head.shape[1] - 5
    counts = np.zeros(4, np.int32)
    o = dict(cand_anchor=np.zeros(cap, np.int32), cand_xywh=np.zeros((cap, 4)), cand_conf=np.zeros(cap),
             cand_cls=np.zeros(cap, np.int32), keep=np.zeros(cap, np.int32), det_xywh=np.zeros((cap, 4)),
             det_conf=np.zeros(cap), det_cls=np.zeros(cap, np.int32), det_xyxy_i=np.zeros((cap, 4), np.int32),
             det_xyxy_d=np.zeros((cap, 4)))
    lib().emu_yolo_post(_p(head), layout, A, nc, C.c_double(box_score), C.c_double(iou), nms_mode,
                        int(lb["pad"][0]), int(lb["pad"][1]), C.c_double(lb["ratio"][0]), C.c_double(lb["ratio"][1]),
                        cap, int(input_hw[0]), int(input_hw[1]), _p(counts), *[_p(o[k]) for k in ("cand_anchor", "cand_xywh", "cand_conf", "cand_cls",
                                                               "keep", "det_xywh", "det_conf", "det_cls",
                                                               "det_xyxy_i", "det_xyxy_d")])
    n, k = int(counts[1]), int(counts[2])
    return dict(n_found=int(counts[0]), overflow=bool(counts[3] & 1),
                cand_anchor=o["cand_anchor"][:n], cand_xywh=o["cand_xywh"][:n], cand_conf=o["cand_conf"][:n],
                cand_cls=o["cand_cls"][:n], keep=o["keep"][:k], xywh=o["det_xywh"][:k], conf=o["det_conf"][:k],
                class_id=o["det_cls"][:k], xyxy_int=o["det_xyxy_i"][:k], xyxy_d=o["det_xyxy_d"][:k])


def ufld(outs, cfg, W, H, lw=1):
    lr, lc, er, ec = [np.ascontiguousarray(o, np.float32) for o in outs]
    cnt = np.zeros(4, np.int32); det = np.zeros(4, np.int32); pts = np.zeros((4, 128, 2), np.int32)
    ra = np.ascontiguousarray(cfg.row_anchor, np.float64); ca = np.ascontiguousarray(cfg.col_anchor, np.float64)
    lib().emu_ufld(_p(lr), _p(lc), _p(er), _p(ec), lr.shape[1], lr.shape[2], lc.shape[1], lc.shape[2], W, H, lw,
                   _p(ra), _p(ca), _p(cnt), _p(det), _p(pts), int(lr.shape[3]))
    return [[(int(x), int(y)) for x, y in pts[i, :cnt[i]]] for i in range(4)], [bool(d) for d in det]


def effdet(boxes, ids, confs, lb, box_score, cap=256):
    import ctypes as C
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4); ids = np.ascontiguousarray(ids, np.int32); confs = np.ascontiguousarray(confs, np.float32)
    n = len(confs)
    cnt = np.zeros(1, np.int32); xywh = np.zeros((cap, 4), np.float32); conf = np.zeros(cap, np.float32)
    cls = np.zeros(cap, np.int32); xi = np.zeros((cap, 4), np.int32)
    lib().emu_effdet(_p(boxes), _p(ids), _p(confs), n, int(lb["pad"][0]), int(lb["pad"][1]), C.c_double(lb["ratio"][0]), C.c_double(lb["ratio"][1]),
                     C.c_double(box_score), cap, _p(cnt), _p(xywh), _p(conf), _p(cls), _p(xi))
    k = int(cnt[0])
    return dict(xywh=xywh[:k], conf=conf[:k], class_id=cls[:k].astype(np.int64), xyxy_int=xi[:k].astype(np.int64))


def effdet_tail(reg, cls, in_hw, score_thr=0.05, iou_thr=0.5, max_det=100, cap=2048, anchor_scale=4.0):
    import ctypes as C
    reg = np.ascontiguousarray(reg, np.float32).reshape(-1, 4); cls = np.ascontiguousarray(cls, np.float32)
    cnt = np.zeros(2, np.int32); boxes = np.zeros((max_det, 4), np.float32); ids = np.zeros(max_det, np.int32); confs = np.zeros(max_det, np.float32)
    lib().emu_effdet_tail(_p(reg), _p(cls), int(in_hw[0]), int(in_hw[1]), int(cls.shape[1]), cap, max_det, C.c_double(score_thr), C.c_double(iou_thr),
                          C.c_double(anchor_scale), _p(cnt), _p(boxes), _p(ids), _p(confs))
    k = int(cnt[0])
    return dict(boxes=boxes[:k], class_id=ids[:k].astype(np.int64), conf=confs[:k], n_candidates=int(cnt[1]))


def ufld1(head, cfg, input_wh, src_wh):
    out = np.ascontiguousarray(head, np.float32)
    cnt = np.zeros(4, np.int32); det = np.zeros(4, np.int32); pts = np.zeros((4, 128, 2), np.int32)
    ra = np.ascontiguousarray(cfg.row_anchor, np.float64)
    lib().emu_ufld1(_p(out), cfg.griding_num, cfg.cls_num_per_lane, cfg.img_w, cfg.img_h, input_wh[0], input_wh[1],
                    src_wh[0], src_wh[1], _p(ra), _p(cnt), _p(det), _p(pts))
    return [[(int(x), int(y)) for x, y in pts[i, :cnt[i]]] for i in range(4)], [bool(d) for d in det]


def lane_geometry(lanes, status, img_h, bird_wh, M, adjust=True):
    """lanes: 4 lists of (x, y); status: 4 bools (a lane decoder's output)."""
    cnt = np.asarray([len(l) for l in lanes], np.int32); det = np.asarray([1 if s else 0 for s in status], np.int32)
    pts = np.zeros((4, 128, 2), np.int32)
    for i, l in enumerate(lanes):
        if len(l):
            pts[i, :len(l)] = np.asarray(l, np.int32)
    hdr = np.zeros(8, np.int32); vals = np.zeros(2); area = np.zeros((2 * img_h, 2), np.int32); bird = np.zeros((4, 128, 2), np.int32)
    m = np.ascontiguousarray(M, np.float64).reshape(9)
    lib().emu_lane_geometry(_p(cnt), _p(det), _p(pts), int(img_h), int(bird_wh[0]), int(bird_wh[1]), 1 if adjust else 0, _p(m),
                            _p(hdr), _p(vals), _p(area), _p(bird))
    n = int(hdr[1] + hdr[2])
    return dict(area_status=bool(hdr[0]), area_points=area[:n].copy(), n_left=int(hdr[1]), n_right=int(hdr[2]),
                bird_points=[bird[i, :cnt[i]].copy() for i in range(4)], direction=(None, "L", "R", "F")[hdr[3]],
                curvature=float(vals[0]) if hdr[3] else None, offset=float(vals[1]) if hdr[3] else None)


class Tracker:
    def __init__(self, track_thresh=0.5, match_thresh=0.8, track_buffer=30, frame_rate=30, MT=256, MD=256):
        self.h = lib().emu_bt_create(track_thresh, match_thresh, track_buffer, frame_rate, MT, MD)
        self.MT = MT

    def reset(self):
        lib().emu_bt_reset(self.h)

    def update(self, boxes, scores, cls):
        b = np.ascontiguousarray(np.asarray(boxes, np.float64).reshape(-1, 4))
        s = np.ascontiguousarray(np.asarray(scores, np.float64).reshape(-1))
        c = np.ascontiguousarray(np.asarray(cls, np.int32).reshape(-1))
        err = lib().emu_bt_update(self.h, _p(b), _p(s), _p(c), len(s))
        hdr = np.zeros(5, np.int32); recs = np.zeros(2 * self.MT, BTOUT)
        lib().emu_bt_fetch(self.h, _p(hdr), _p(recs))
        return snapshot(hdr, recs), err

    def trajectories(self):
        """STrack.trajectories of the tracks in message order (tracked, then lost): list of (len, 4) arrays."""
        hdr = np.zeros(5, np.int32); recs = np.zeros(2 * self.MT, BTOUT)
        lib().emu_bt_fetch(self.h, _p(hdr), _p(recs))
        n = int(hdr[2] + hdr[3])
        lens = np.zeros(max(n, 1), np.int32); out = np.zeros((max(n, 1), 30, 4))
        lib().emu_bt_trajectories(self.h, _p(lens), _p(out))
        assert [int(v) for v in lens[:n]] == [int(r["traj_len"]) for r in recs[:n]]
        return [out[k, :lens[k]].copy() for k in range(n)]

    def __del__(self):
        try:
            lib().emu_bt_destroy(self.h)
        except Exception:
            pass


def snapshot(hdr, recs):
    nt, nl = int(hdr[2]), int(hdr[3])

    def rec(r):
        return dict(track_id=int(r["track_id"]), state=int(r["state"]), is_activated=bool(r["is_activated"]),
                    score=float(r["score"]), class_id=int(r["class_id"]), start_frame=int(r["start_frame"]),
                    frame_id=int(r["frame_id"]), tracklet_len=int(r["tracklet_len"]),
                    tlwh=[float(v) for v in r["tlwh"]])
    return dict(frame_id=int(hdr[0]), count=int(hdr[1]), tracked=[rec(r) for r in recs[:nt]],
                lost=[rec(r) for r in recs[nt:nt + nl]])
