"""CPU: the device post-processing *logic* (csrc/post_core.h, track_core.h compiled for the host,
single thread) against the oracle and the reference goldens.  The real HIP kernels are checked by
tests/test_gpu_*.py with the same assertions."""
import gzip, json, os
import numpy as np
import pytest

from conftest import GOLDEN
import synth, emu_api, parity_checks as pc
from oracle import yolo_post, ufld_decode, bytetrack

MT = {"YOLOV8": ("yolov8", 0), "YOLOV5": ("yolov5", 1)}


@pytest.mark.parametrize("case", synth.yolo_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("mode", [0, 1])
def test_yolo_post(case, mode):
    tag, mt, head, lb, bs, iou = case
    name, layout = MT[mt]
    lbp = yolo_post.letterbox_params(lb["old"], lb["target"])
    want = yolo_post.detect_post(head, lbp, name, bs, iou, "reference" if mode == 0 else "greedy")
    got = emu_api.yolo_post(head, layout, lbp, bs, iou, mode)
    assert not got["overflow"]
    pc.check_yolo(got, want)
    if mode == 0:  # and directly against the reference's own output
        g = np.load(os.path.join(GOLDEN, "yolo_post.npz"))
        np.testing.assert_array_equal(got["keep"], g[tag + "_keep"])
        np.testing.assert_array_equal(got["xyxy_int"], g[tag + "_rect_xyxy_int"])


@pytest.mark.parametrize("case", synth.lite_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("mode", [0, 1])
def test_yolo_lite_post(case, mode):
    """v5-lite grid decode inside the candidate stage (layout 2) vs the oracle and the reference's own run."""
    tag, head, hw, lb, bs, iou = case
    lbp = yolo_post.letterbox_params(lb["old"], lb["target"])
    want = yolo_post.detect_post(head, lbp, "yolov5_lite", bs, iou, "reference" if mode == 0 else "greedy", input_hw=hw)
    got = emu_api.yolo_post(head, 2, lbp, bs, iou, mode, input_hw=hw)
    assert not got["overflow"]
    pc.check_yolo(got, want)
    if mode == 0:
        g = np.load(os.path.join(GOLDEN, "yolo_lite.npz"))
        np.testing.assert_array_equal(got["keep"], g[tag + "_keep"])
        np.testing.assert_array_equal(got["xyxy_int"], g[tag + "_rect_xyxy_int"])
        np.testing.assert_array_equal(got["cand_xywh"], g[tag + "_xywh"])


def test_yolo_post_generic_v5_product():
    """v5 conf = cls*obj rounded in fp32 (yoloDetector.py:124) on non-dyadic values."""
    rng = np.random.default_rng(5)
    head = rng.uniform(0, 1, (25200, 85)).astype(np.float32)
    head[:, :4] = rng.uniform(20, 600, (25200, 4))
    head[:, 4] *= (rng.uniform(0, 1, 25200) < 0.004)
    lbp = yolo_post.letterbox_params((720, 1280), (640, 640))
    want = yolo_post.detect_post(head, lbp, "yolov5", 0.4, 0.45)
    got = emu_api.yolo_post(head, 1, lbp, 0.4, 0.45, 0)
    assert len(want["keep"]) > 5
    pc.check_yolo(got, want)


def test_yolo_post_overflow_flag():
    head = synth.synth_v8_head(4, 600, 200)
    lbp = yolo_post.letterbox_params((720, 1280), (640, 640))
    got = emu_api.yolo_post(head, 0, lbp, 0.4, 0.45, 0, cap=128)
    assert got["overflow"] and got["n_found"] > 128 and len(got["cand_anchor"]) == 128


def test_nms_kats_via_head():
    """SURVEY KATs 1-4 pushed through the whole device chain (square letterbox = identity)."""
    lbp = yolo_post.letterbox_params((640, 640), (640, 640))
    kats = [([(0, 0, 10, 10), (100, 100, 10, 10), (200, 200, 10, 10)], [.5, .9, .7], [1, 2]),
            ([(0, 0, 100, 100), (300, 300, 10, 10), (500, 500, 10, 10)], [.5, .9, .7], [1, 2, 2]),
            ([(0, 0, 10, 10), (1, 1, 10, 10), (50, 50, 10, 10)], [.9, .8, .7], [0, 2])]
    for boxes, scores, keep in kats:
        head = np.zeros((84, 8400), np.float32)
        for i, ((x, y, w, h), s) in enumerate(zip(boxes, scores)):
            a = 100 * (i + 1)
            head[0:4, a] = [x + w / 2, y + h / 2, w, h]
            head[4 + i, a] = s
        got = emu_api.yolo_post(head, 0, lbp, 0.4, 0.45, 0)
        assert got["keep"].tolist() == keep


@pytest.mark.parametrize("case", synth.ufld_cases(), ids=lambda c: c[0])
def test_ufld(case):
    tag, outs, W, H = case
    cfg = ufld_decode.ModelConfig("culane")
    want_l, want_s = ufld_decode.process_output(outs, cfg, W, H)
    got_l, got_s = emu_api.ufld(outs, cfg, W, H)
    pc.check_lanes(got_l, got_s, want_l, want_s, tol_px=0)


@pytest.mark.parametrize("case", synth.curve_cases(), ids=lambda c: c[0])
def test_ufld_curvelanes_ten_lane_heads(case):
    """The decoder logic on 10-lane tensors (CurveLanes configuration): lane stride 10, lanes 1,2 / 0,3 decoded."""
    tag, outs, W, H = case
    cfg = ufld_decode.ModelConfig("curvelanes")
    want_l, want_s = ufld_decode.process_output(outs, cfg, W, H)
    got_l, got_s = emu_api.ufld(outs, cfg, W, H)
    pc.check_lanes(got_l, got_s, want_l, want_s, tol_px=1)


@pytest.mark.parametrize("case", synth.effdet_cases(), ids=lambda c: c[0])
def test_effdet_post(case):
    from oracle import effdet_post
    tag, boxes, ids, confs, src, inp, thr = case
    lb = yolo_post.letterbox_params(src, inp)
    want = effdet_post.process_output(boxes, ids, confs, lb, thr)
    got = emu_api.effdet(boxes, ids, confs, lb, thr)
    for k in ("xywh", "conf", "class_id", "xyxy_int"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


@pytest.mark.parametrize("case", synth.ufld1_cases(), ids=lambda c: c[0])
def test_ufld_v1(case):
    """UFLD v1 decode logic (host build; exp in libm double then rounded, NumPy's is fp32 SIMD): +-1 px."""
    tag, cfgname, head, iwh, swh = case
    cfg = ufld_decode.ModelConfigV1(cfgname)
    want_l, want_s = ufld_decode.process_output_v1(head, cfg, iwh[0], iwh[1], swh[0], swh[1])
    got_l, got_s = emu_api.ufld1(head, cfg, iwh, swh)
    n_off = pc.check_lanes(got_l, got_s, want_l, want_s, tol_px=1)
    assert n_off <= 2


@pytest.mark.parametrize("tag", ["t1", "t2", "t3", "t4", "t5", "t6", "t7"])
def test_bytetrack(tag):
    with gzip.open(os.path.join(GOLDEN, "bytetrack.json.gz"), "rt") as f:
        sc = json.load(f)[tag]
    trk = emu_api.Tracker()
    for fr, want in zip(sc["frames"], sc["trace"]):
        got, err = trk.update(fr["boxes"], fr["scores"], fr["ids"])
        assert err == 0
        if sc["label_ids"]:
            lab = ["car", "person", "truck"]
            for lst in ("tracked", "lost"):
                for t in got[lst]:
                    t["class_id"] = lab[t["class_id"]]
        pc.check_track_frame(got, want, ctx=(tag, want["frame_id"]))


@pytest.mark.parametrize("tag", ["t1", "t2", "t3", "t4", "t5", "t6", "t7"])
def test_bytetrack_trajectories(tag):
    """The device tracker's trajectory rings (track_core.h bt_apply_match / bytetrack_gather_trajectories, compiled for the host) vs
    STrack.trajectories of the REFERENCE tracker (tests/golden/make_golden_traj.py) at its checkpoints: ids, lengths, every box exact."""
    with gzip.open(os.path.join(GOLDEN, "bytetrack.json.gz"), "rt") as f:
        sc = json.load(f)[tag]
    with gzip.open(os.path.join(GOLDEN, "bytetrack_traj.json.gz"), "rt") as f:
        want = json.load(f)[tag]
    trk = emu_api.Tracker()
    seen = 0
    for k, fr in enumerate(sc["frames"]):
        got, err = trk.update(fr["boxes"], fr["scores"], fr["ids"])
        assert err == 0
        if str(k) not in want:
            continue
        seen += 1
        ids = [t["track_id"] for t in got["tracked"] + got["lost"]]
        assert ids == [r["track_id"] for r in want[str(k)]]
        for tr, r in zip(trk.trajectories(), want[str(k)]):
            assert tr.shape == (len(r["trajectory"]), 4) and (len(tr) == 30) == r["full"]
            assert tr.tolist() == r["trajectory"], (tag, k, r["track_id"])
    assert seen == len(want)
    trk.reset()
    got, err = trk.update(sc["frames"][-1]["boxes"], sc["frames"][-1]["scores"], sc["frames"][-1]["ids"])
    assert all(len(t) == 0 for t in trk.trajectories())        # fresh tracks after a reset: empty lists (activate() appends nothing)


def test_bytetrack_reset_and_random_vs_oracle():
    rng = np.random.default_rng(42)
    trk = emu_api.Tracker(); ora = bytetrack.BYTETracker()
    for rep in range(2):
        frames = synth.track_scene(100 + rep, 25, 50, 0.2)
        for f, fr in enumerate(frames):
            got, err = trk.update(fr["boxes"], fr["scores"], fr["ids"])
            want = ora.update(fr["boxes"], fr["scores"], fr["ids"])
            assert err == 0
            pc.check_track_frame(got, want, ctx=(rep, f))
        trk.reset(); ora.reset()


def test_bytetrack_large_scene_logic():
    """Same scenario family as the GPU test of the register-resident solver (here the block-wide form, one thread)."""
    frames = synth.track_scene(400, 100, 8, 0.08, 1920, 1080)
    trk = emu_api.Tracker(MT=512, MD=256)
    ora = bytetrack.BYTETracker()
    for f, fr in enumerate(frames):
        got, err = trk.update(fr["boxes"], fr["scores"], fr["ids"])
        assert err == 0
        pc.check_track_frame(got, ora.update(fr["boxes"], fr["scores"], fr["ids"]), ctx=f)


# ------------------------------------------------------------------------------------------------ lane geometry (f2)
def _geometry_reference(lanes, status, W, H, M, adjust=True):
    """The host mirror of the reference's lane geometry (analysis.py / oracle), pinned by tests/test_analysis.py and
    tests/test_oracle_golden.py against the reference's own runs."""
    import importlib
    from conftest import load_pkg
    load_pkg()
    A = importlib.import_module("adas_amd.analysis")
    st, area = ufld_decode.lanes_area(lanes, status, H, adjust=adjust)
    tv = A.PerspectiveTransformation((W, H))
    tv.M = np.asarray(M, np.float64).reshape(3, 3)
    bird = [np.asarray(tv.transformToBirdViewPoints(l), np.int64).reshape(-1, 2) for l in lanes]
    try:
        (d, cv), off = tv.calcCurveAndOffset((H, W), bird[1], bird[2])
    except IndexError:          # bird view lower than 720 rows: the reference reads row 719 (perspectiveTransformation.py:196)
        d, cv, off = None, None, None      # the device routine reports "no estimate" there
    return st, np.asarray(area, np.int64).reshape(-1, 2), bird, d, cv, off


def check_geometry(got, want, max_off_points=2):
    st, area, bird, d, cv, off = want
    assert got["area_status"] == st
    assert got["area_points"].shape == area.shape
    diff = np.abs(got["area_points"].astype(np.int64) - area)
    assert diff.max(initial=0) <= 1 and int((diff > 0).sum()) <= max_off_points     # QR vs LAPACK SVD at an integer boundary
    for i in range(4):
        np.testing.assert_array_equal(got["bird_points"][i], bird[i])
    assert got["direction"] == d
    if d is not None:
        assert got["curvature"] == pytest.approx(cv, rel=1e-7) and got["offset"] == pytest.approx(off, rel=1e-7, abs=1e-9)


@pytest.mark.parametrize("case", synth.ufld_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("adjust", [True, False])
def test_lane_geometry_on_decoded_lanes(case, adjust):
    import importlib
    from conftest import load_pkg
    load_pkg()
    A = importlib.import_module("adas_amd.analysis")
    tag, outs, W, H = case
    lanes, status = ufld_decode.process_output(outs, ufld_decode.ModelConfig("culane"), W, H)
    M = A.PerspectiveTransformation((W, H)).M
    got = emu_api.lane_geometry(lanes, status, H, (W, H), M, adjust)
    check_geometry(got, _geometry_reference(lanes, status, W, H, M, adjust))


def test_lane_geometry_reference_goldens():
    """Bird-view points, direction, curvature and offset of the reference's own PerspectiveTransformation runs
    (tests/golden/analysis.json.gz), through the device routine."""
    with gzip.open(os.path.join(GOLDEN, "analysis.json.gz"), "rt") as f:
        g = json.load(f)["perspective"]
    lanes = [[], [tuple(p) for p in g["left"]], [tuple(p) for p in g["right"]], []]
    for st in g["steps"]:
        got = emu_api.lane_geometry(lanes, [False, True, True, False], 720, (1280, 720), st["M"], True)
        np.testing.assert_array_equal(got["bird_points"][1], np.array(st["bird_left"]))
        np.testing.assert_array_equal(got["bird_points"][2], np.array(st["bird_right"]))
        assert got["direction"] == st["direction"]
        assert got["curvature"] == pytest.approx(st["curvature"], rel=1e-7) and got["offset"] == pytest.approx(st["offset"], rel=1e-7, abs=1e-9)
    cv = g["curvy"]   # bird-view lanes given directly: identity homography
    lanes = [[], [tuple(p) for p in cv["left"]], [tuple(p) for p in cv["right"]], []]
    got = emu_api.lane_geometry(lanes, [False, True, True, False], 720, (1280, 720), np.eye(3), False)
    assert got["direction"] == cv["direction"] and got["curvature"] == pytest.approx(cv["curvature"], rel=1e-7)
    assert got["offset"] == pytest.approx(cv["offset"], rel=1e-7)
    empty = emu_api.lane_geometry([[], [], [], []], [False] * 4, 720, (1280, 720), np.eye(3), True)
    assert empty["direction"] is None and not empty["area_status"] and len(empty["area_points"]) == 0


def _effdet_heads(seed, in_hw, nc=90, bias=-8.0, spread=1.0, n_obj=12):
    """Synthetic raw head tensors of one frame: background logits around `bias`, a few object blobs (neighbouring anchors of one class
    with correlated regressions, so the NMS has overlapping same-class boxes to suppress and overlapping other-class boxes to keep)."""
    rng = np.random.default_rng(seed)
    A = 9 * sum((in_hw[0] >> l) * (in_hw[1] >> l) for l in range(3, 8))
    cls = (bias + spread * rng.standard_normal((A, nc))).astype(np.float32)
    reg = (0.3 * rng.standard_normal((A, 4))).astype(np.float32)
    for _ in range(n_obj):
        a0 = int(rng.integers(0, A - 40)); c = int(rng.integers(0, nc))
        idx = a0 + rng.choice(40, 12, replace=False)
        cls[idx, c] = rng.uniform(-1.0, 3.0, 12).astype(np.float32)
        cls[idx[:3], (c + 1) % nc] = rng.uniform(0.0, 3.0, 3).astype(np.float32)
        reg[idx] = (0.05 * rng.standard_normal((12, 4))).astype(np.float32)
    return reg, cls


@pytest.mark.parametrize("seed,in_hw,thr,iou,max_det", [(0, (128, 128), 0.05, 0.5, 100), (1, (256, 384), 0.2, 0.5, 100), (2, (128, 256), 0.05, 0.3, 7),
                                                        (3, (512, 512), 0.3, 0.5, 100), (4, (128, 128), 0.999, 0.5, 100)])
def test_effdet_tail(seed, in_hw, thr, iou, max_det):
    """The in-graph tail of EfficientDet (csrc/post_core.h effdet_tail_frame, host build) against the numpy restatement: candidates,
    order, decoded + clipped boxes, per-class suppression, the max_det cut -- bit for bit."""
    from oracle import effdet_tail
    reg, cls = _effdet_heads(seed, in_hw)
    want = effdet_tail.tail(reg, cls, in_hw, thr, iou, max_det)
    got = emu_api.effdet_tail(reg, cls, in_hw, thr, iou, max_det, cap=3072)
    assert got["n_candidates"] == want["n_candidates"] <= 3072
    if thr < 0.9:
        assert len(want["conf"]) >= min(max_det, 5) and want["n_candidates"] > len(want["conf"])      # the NMS has something to do
    else:
        assert want["n_candidates"] == 0 and len(got["conf"]) == 0
    np.testing.assert_array_equal(got["class_id"], want["class_id"])
    np.testing.assert_array_equal(got["conf"], want["conf"])
    np.testing.assert_array_equal(got["boxes"], want["boxes"])


def test_effdet_anchor_table():
    """oracle.effdet_tail.anchors: counts, order and the D0 geometry the paper states (anchor side 4 x stride x 2^(k/3), three aspect ratios)."""
    from oracle import effdet_tail
    a = effdet_tail.anchors(512, 512)
    assert a.shape == (49104, 4)                                  # 9 * (64^2 + 32^2 + 16^2 + 8^2 + 4^2)
    np.testing.assert_allclose(a[0], [4 - 16, 4 - 16, 4 + 16, 4 + 16])                 # level 3, cell (0, 0), scale 1, ratio (1, 1): side 32
    np.testing.assert_allclose(a[1], [4 - 0.7 * 16, 4 - 1.4 * 16, 4 + 0.7 * 16, 4 + 1.4 * 16], rtol=1e-6)
    np.testing.assert_allclose(a[9], [4 - 16, 12 - 16, 4 + 16, 12 + 16])               # next cell along x
    last = a[-1]                                                  # level 7, last cell, scale 2^(2/3), ratio (0.7, 1.4)
    side = 4 * 128 * 2 ** (2 / 3)
    np.testing.assert_allclose(last, [448 - 1.4 * side / 2, 448 - 0.7 * side / 2, 448 + 1.4 * side / 2, 448 + 0.7 * side / 2], rtol=1e-6)


def test_effdet_tail_overflow_is_reported():
    """More anchors over the score threshold than max_candidates: the count is reported (the C-ABI turns it into ADAS_ERR_CAPACITY), nothing is kept."""
    reg, cls = _effdet_heads(5, (128, 128), bias=0.0)
    got = emu_api.effdet_tail(reg, cls, (128, 128), 0.05, 0.5, 100, cap=64)
    assert got["n_candidates"] > 64 and len(got["conf"]) == 0
