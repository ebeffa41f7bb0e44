#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python under import stubs.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):   python tests/golden/make_golden.py
Writes tests/golden/{nms_kat,yolo_post,ufld_decode}.npz and bytetrack.json.gz
(inputs come from tests/golden/synth.py; fixtures hold outputs + input digests).

Stubs: numba.jit = identity decorator, cv2 = constants
only, empty onnxruntime/tensorrt/pycuda, lap.lapjv = SciPy LSA on lap's
extended matrix, np.float = float.  No reference source is copied: the
reference modules are imported from where they lie and their functions called.

Pinned-env promotion (SURVEY finding 5): v8 heads are fed float64-widened so
that yoloDetector.py:132 runs in fp64 as it does under numpy==1.22.1; v5 heads
use dyadic-rational obj/cls values so the fp32 product at :124 is exact and the
widened flow is bit-identical to the pinned one.
"""
import json, os, sys, tempfile, types
import numpy as np

REF = os.environ.get("ADAS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    d = tempfile.mkdtemp(prefix="adas_stubs_")
    open(os.path.join(d, "numba.py"), "w").write(
        "def jit(*a, **k):\n    def deco(f):\n        return f\n    return deco\n")
    open(os.path.join(d, "cv2.py"), "w").write(
        "INTER_LINEAR=1\nFONT_HERSHEY_TRIPLEX=4\nLINE_AA=16\nCOLOR_BGR2RGB=4\nFONT_HERSHEY_SIMPLEX=0\n")
    for m in ("onnxruntime", "tensorrt"):
        open(os.path.join(d, m + ".py"), "w").write("")
    os.makedirs(os.path.join(d, "pycuda"))
    open(os.path.join(d, "pycuda", "__init__.py"), "w").write("")
    open(os.path.join(d, "pycuda", "driver.py"), "w").write("")
    open(os.path.join(d, "lap.py"), "w").write(
        "import numpy as np\nfrom scipy.optimize import linear_sum_assignment\n"
        "def lapjv(cost, extend_cost=False, cost_limit=np.inf):\n"
        "    T, D = cost.shape\n    n = T + D\n"
        "    e = np.full((n, n), cost_limit / 2.0)\n    e[T:, D:] = 0\n    e[:T, :D] = cost\n"
        "    r, c = linear_sum_assignment(e)\n"
        "    x = np.full(n, -1, dtype=int); y = np.full(n, -1, dtype=int)\n"
        "    x[r] = c; y[c] = r\n    x = x[:T].copy(); y = y[:D].copy()\n"
        "    x[x >= D] = -1; y[y >= T] = -1\n"
        "    return float(e[r, c].sum()), x, y\n")
    sys.path.insert(0, d)
    sys.path.insert(1, REF)
    np.float = float  # removed in NumPy >= 1.24; matching.py:72,75,76, strack.py:40


sys.path.insert(0, HERE)
from synth import *  # noqa: E402,F401  (seeded inputs, no reference dependency)


# ------------------------------------------------------------------ reference runners
def ref_yolo_chain(head, model_type_name, lb, box_score, iou):
    from ObjectDetector.yoloDetector import YoloDetector
    from ObjectDetector.utils import ObjectModelType, Scaler, NMS
    det = object.__new__(YoloDetector)
    det.model_type = getattr(ObjectModelType, model_type_name)
    det.box_score = box_score
    det.box_nms_iou = iou
    det.lite = False
    det.class_names = [str(i) for i in range(80)]
    boxes, cids, confs, _ = det._YoloDetector__process_output(head.astype(np.float64))
    sc = Scaler(lb["target"], True)
    sc._old_shape, sc._new_shape, sc._pad_shape = lb["old"], lb["new"], lb["pad"]
    tb = sc.convert_boxes_coordinate(boxes)
    tb = np.asarray(tb, np.float64).reshape(-1, 4)
    keep = NMS.fast_soft_nms(tb, confs, iou, dets_type="xywh")
    keep_alt = NMS.fast_nms(tb, confs, iou, "xywh")
    infos = det.get_nms_results(tb, confs, cids, np.array([]))
    return dict(
        raw_boxes=np.asarray(boxes, np.float64).reshape(-1, 4), cls=np.asarray(cids, np.int64),
        conf=np.asarray(confs, np.float64), xywh=tb, keep=np.asarray(keep, np.int64),
        keep_alt=np.asarray(keep_alt, np.int64),
        rect_xywh=np.asarray([[r.x, r.y, r.width, r.height] for r in infos], np.float64).reshape(-1, 4),
        rect_conf=np.asarray([r.conf for r in infos], np.float64),
        rect_label=np.asarray([int(r.label) for r in infos], np.int64),
        rect_xyxy_int=np.asarray([r.tolist() for r in infos], np.int64).reshape(-1, 4))


def ref_ufld(outputs, W, H, adjust=True):
    from TrafficLaneDetector.ufldDetector.ultrafastLaneDetectorV2 import UltrafastLaneDetectorV2, ModelConfig
    from TrafficLaneDetector.ufldDetector.utils import LaneModelType
    from TrafficLaneDetector.ufldDetector.core import LaneInfo
    d = object.__new__(UltrafastLaneDetectorV2)
    d.cfg = ModelConfig(LaneModelType.UFLDV2_CULANE)
    d.img_width, d.img_height = W, H
    d.lane_info = LaneInfo(np.array([], dtype=object), np.array([], dtype=object), np.array([], dtype=object), False)
    d.adjust_lanes = adjust
    pts, status = d._UltrafastLaneDetectorV2__process_output(outputs, d.cfg)
    d.lane_info.lanes_points, d.lane_info.lanes_status = pts, status
    d._LaneDetectBase__update_lanes_status(status)
    d._LaneDetectBase__update_lanes_area(pts, H)
    lanes = [[(int(p[0]), int(p[1])) for p in lane] for lane in pts]
    area = np.asarray(d.lane_info.area_points, dtype=np.int64).reshape(-1, 2) if d.lane_info.area_status else np.zeros((0, 2), np.int64)
    return lanes, [bool(s) for s in status], bool(d.lane_info.area_status), area


def ref_track_run(frames, label_ids=False):
    from ObjectTracker.byteTrack.byteTracker import BYTETracker
    from ObjectTracker.byteTrack.dtypes import BaseTrack, STrack
    BaseTrack.reset_counter()
    STrack.update_crops = lambda self, frame: None       # pixel crops: side data, needs the host frame
    names = {"car": (0, 0, 255), "person": (0, 255, 0), "truck": (255, 0, 0)}
    lab = list(names.keys())
    trk = BYTETracker(names=names)

    def rec(t):
        cid = t.class_id
        return dict(track_id=int(t.track_id), state=int(t.state), is_activated=bool(t.is_activated),
                    score=float(t.score), class_id=(cid if isinstance(cid, str) else int(cid)),
                    start_frame=int(t.start_frame), frame_id=int(t.frame_id),
                    tracklet_len=int(t.tracklet_len), tlwh=[float(v) for v in t.tlwh])
    trace = []
    for fr in frames:
        ids = [lab[i] for i in fr["ids"]] if label_ids else fr["ids"]
        trk.update(fr["boxes"], fr["scores"], ids, None)
        trace.append(dict(frame_id=int(trk.frame_id), count=int(BaseTrack._count),
                          tracked=[rec(t) for t in trk.tracked_stracks],
                          lost=[rec(t) for t in trk.lost_stracks]))
    return trace


# ------------------------------------------------------------------ main
def main():
    install_stubs()
    from ObjectDetector.utils import NMS

    # ---- NMS KATs + random clustered sets
    nms = {}
    kats = [
        ([(0, 0, 10, 10), (100, 100, 10, 10), (200, 200, 10, 10)], [.5, .9, .7]),
        ([(0, 0, 10, 10), (100, 100, 10, 10), (200, 200, 10, 10)], [.9, .7, .5]),
        ([(0, 0, 100, 100), (300, 300, 10, 10), (500, 500, 10, 10)], [.5, .9, .7]),
        ([(0, 0, 10, 10), (1, 1, 10, 10), (50, 50, 10, 10)], [.9, .8, .7]),
        ([(5, 5, 20, 20)], [.8]),
    ]
    rng = np.random.default_rng(7)
    for n in (2, 7, 33, 64, 65, 130, 300):
        nc = max(1, n // 4)
        ctr = rng.uniform(50, 1200, (nc, 2)); wh = rng.uniform(20, 200, (nc, 2))
        own = rng.integers(0, nc, n)
        xy = ctr[own] + rng.normal(0, 6, (n, 2)); w_h = wh[own] * rng.uniform(0.85, 1.15, (n, 2))
        boxes = np.concatenate([xy - w_h / 2, w_h], 1)
        scores = rng.uniform(0.4, 0.99, n).astype(np.float32).astype(np.float64)
        if n >= 33:
            scores[5] = scores[11]          # exact score ties
            scores[20] = scores[3]
        kats.append((boxes.tolist(), scores.tolist()))
    for i, (b, s) in enumerate(kats):
        b = np.asarray(b, np.float64).reshape(-1, 4); s = np.asarray(s, np.float64)
        for thr in (0.45, 0.3):
            k1 = np.asarray(NMS.fast_soft_nms(b, s.tolist(), thr, dets_type="xywh"), np.int64)
            k2 = np.asarray(NMS.fast_nms(b, s.tolist(), thr, "xywh"), np.int64)
            tag = f"k{i}_t{int(thr * 100)}"
            nms[tag + "_boxes"] = b; nms[tag + "_scores"] = s
            nms[tag + "_keep"] = k1; nms[tag + "_keep_alt"] = k2
    nms["n_cases"] = np.int64(len(kats))
    np.savez_compressed(os.path.join(HERE, "nms_kat.npz"), **nms)
    print("nms_kat: KAT1", nms["k0_t45_keep"], "KAT3", nms["k2_t45_keep"])

    # ---- YOLO post chain
    yp = {}
    cases = yolo_cases()
    for tag, mt, head, lb, bs, iou in cases:
        r = ref_yolo_chain(head, mt, lb, bs, iou)
        yp[tag + "_head_sha1"] = np.asarray(digest(head))
        yp[tag + "_lb"] = np.asarray([*lb["old"], *lb["new"], *lb["pad"], *lb["target"]], np.int64)
        yp[tag + "_thr"] = np.asarray([bs, iou], np.float64)
        for k, v in r.items():
            yp[f"{tag}_{k}"] = v
        print(tag, "cands", len(r["conf"]), "keep", len(r["keep"]), "alt", len(r["keep_alt"]))
    yp["tags"] = np.asarray([c[0] for c in cases])
    yp["types"] = np.asarray([c[1] for c in cases])
    np.savez_compressed(os.path.join(HERE, "yolo_post.npz"), **yp)

    # ---- UFLD decode
    uf = {}
    ucases = ufld_cases()
    for tag, outs, W, H in ucases:
        lanes, status, astat, area = ref_ufld(outs, W, H)
        uf[f"{tag}_in_sha1"] = np.asarray(digest(*outs))
        uf[f"{tag}_wh"] = np.asarray([W, H], np.int64)
        for li, lane in enumerate(lanes):
            uf[f"{tag}_lane{li}"] = np.asarray(lane, np.int64).reshape(-1, 2)
        uf[f"{tag}_status"] = np.asarray(status, np.bool_)
        uf[f"{tag}_area_status"] = np.bool_(astat)
        uf[f"{tag}_area"] = area
        print(tag, status, [len(l) for l in lanes], "area", astat, area.shape)
    uf["tags"] = np.asarray([c[0] for c in ucases])
    np.savez_compressed(os.path.join(HERE, "ufld_decode.npz"), **uf)

    # ---- ByteTrack traces
    bt = {}
    rng0 = np.random.default_rng(0)
    base = np.array([[100, 100, 200, 220], [400, 300, 520, 380], [800, 200, 860, 330]], float)
    t1 = []
    for f in range(6):
        b = base + f * np.array([5, 2, 5, 2]) + rng0.normal(0, 1, (3, 4))
        s = [.9, .8, .3 if f % 2 else .75]
        ids = [0, 0, 1]
        n = 2 if f == 3 else 3
        t1.append(dict(boxes=b[:n].tolist(), scores=s[:n], ids=ids[:n]))
    bt["t1"] = dict(frames=t1, label_ids=True, trace=ref_track_run(t1, True))
    for tag, seed, nobj, nfr, drop in (("t2", 3, 12, 60, 0.10), ("t3", 4, 40, 80, 0.15), ("t4", 5, 3, 90, 0.5),
                                        ("t5", 6, 90, 40, 0.05)):
        fr = track_scene(seed, nobj, nfr, drop)
        bt[tag] = dict(frames=fr, label_ids=False, trace=ref_track_run(fr, False))
    # t6: long occlusion -> age-out (max_time_lost 30) and lingering-removed semantics
    fr = track_scene(8, 6, 120, 0.0)
    for f in range(20, 70):
        fr[f] = dict(boxes=fr[f]["boxes"][:2], scores=fr[f]["scores"][:2], ids=fr[f]["ids"][:2])
    bt["t6"] = dict(frames=fr, label_ids=False, trace=ref_track_run(fr, False))
    # t7: empty frames
    fr = track_scene(9, 5, 30, 0.2)
    fr[0] = dict(boxes=[], scores=[], ids=[]); fr[10] = dict(boxes=[], scores=[], ids=[])
    bt["t7"] = dict(frames=fr, label_ids=False, trace=ref_track_run(fr, False))
    for k, v in bt.items():
        ids = sorted({t["track_id"] for f in v["trace"] for t in f["tracked"] + f["lost"]})
        print(k, "frames", len(v["frames"]), "max id", max(ids) if ids else 0, "final count", v["trace"][-1]["count"])
    import gzip
    with gzip.open(os.path.join(HERE, "bytetrack.json.gz"), "wt") as f:
        json.dump(bt, f)


if __name__ == "__main__":
    main()
