"""GPU: device pre-processing (pre_kernels.hip) against the oracle restatement, bit-exact, and the drop-in
YoloDetector / UltrafastLaneDetectorV2 / BYTETracker classes end to end (frame in, RectInfo / LaneInfo / tracks out)."""
import importlib
import os
import numpy as np
import pytest

import netutil, parity_checks as pc
from conftest import load_pkg, GOLDEN
import synth
from oracle import preprocess, yolo_post, ufld_decode, bytetrack

pytestmark = pytest.mark.gpu
load_pkg()
L = importlib.import_module("adas_amd._lib")
D = importlib.import_module("adas_amd.detectors")
CE = importlib.import_module("adas_amd.coreEngine")


def frames(n, h, w, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    f[:, h // 4:h // 2, w // 3:w // 2] = rng.integers(0, 256, (n, 1, 1, 3), dtype=np.uint8)   # flat patches
    return f


@pytest.mark.parametrize("hw", [(720, 1280), (640, 640), (1080, 1920), (800, 600), (333, 517)], ids=str)
def test_preprocess_yolo_bit_exact(hw):
    f = frames(2, hw[0], hw[1], 3)
    d_in = L.DeviceBuffer.from_array(f)
    d_out = L.DeviceBuffer(2 * 3 * 640 * 640 * 4)
    L.check(L.lib().adas_preprocess_yolo(d_in.ptr, 2, hw[0], hw[1], d_out.ptr, 640, 640, 1, None))
    L.check(L.lib().adas_synchronize())
    got = d_out.download((2, 3, 640, 640), np.float32)
    for i in range(2):
        np.testing.assert_array_equal(got[i], preprocess.yolo_prepare_input(f[i], (640, 640))[0])


@pytest.mark.parametrize("hw", [(720, 1280), (533, 1600), (1080, 1920)], ids=str)
def test_preprocess_ufld_bit_exact(hw):
    f = frames(2, hw[0], hw[1], 4)
    d_in = L.DeviceBuffer.from_array(f)
    d_out = L.DeviceBuffer(2 * 3 * 320 * 1600 * 4)
    L.check(L.lib().adas_preprocess_ufld(d_in.ptr, 2, hw[0], hw[1], d_out.ptr, 320, 1600, 0.6, None))
    L.check(L.lib().adas_synchronize())
    got = d_out.download((2, 3, 320, 1600), np.float32)
    for i in range(2):
        np.testing.assert_array_equal(got[i], preprocess.ufld_prepare_input(f[i], (320, 1600), 0.6)[0])


def calibrated_v8n(tmp_path):
    """yolov8n with the Detect cls bias raised so a handful of anchors pass box_score on noise frames."""
    import bench
    M = importlib.import_module("adas_amd.models")
    x = np.stack([preprocess.yolo_prepare_input(f, (640, 640))[0] for f in frames(2, 720, 1280, 7)])
    path, W, g = bench.build_detector(M, CE, "yolov8n", x, str(tmp_path), "t", target_per_frame=80.0)
    return path


def test_yolo_detector_dropin(tmp_path):
    path = calibrated_v8n(tmp_path)
    lab = tmp_path / "coco_label.txt"
    lab.write_text("\n".join(f"class{i}" for i in range(79)))          # one short: the last id maps to "unknown" (yoloDetector.py:144-147)
    det = D.YoloDetector(model_path=path, model_type=D.ObjectModelType.YOLOV8, classes_path=str(lab), box_score=0.4,
                         box_nms_iou=0.45, precision="fp32")
    eng = CE.OnnxEngine(path, precision="fp32")
    n_total = 0
    for f in frames(3, 720, 1280, 7):
        det.DetectFrame(f)
        # expected: reference post-processing (oracle) applied to the engine's head for the oracle-pre-processed frame
        head = eng.engine_inference(preprocess.yolo_prepare_input(f, (640, 640)))[0][0]
        lb = yolo_post.letterbox_params((720, 1280), (640, 640))
        want = yolo_post.detect_post(head, lb, "yolov8", 0.4, 0.45)
        pc.check_yolo(det._last, want)
        info = det.object_info
        assert len(info) == len(want["keep"])
        for r, xywh, conf, cid, xyxy in zip(info, want["xywh"], want["conf"], want["class_id"], want["xyxy_int"]):
            assert isinstance(r, D.RectInfo) and r.tolist() == list(xyxy) and r.conf == conf
            assert (r.x, r.y, r.width, r.height) == tuple(xywh)
            assert r.label == (f"class{cid}" if cid < 79 else "unknown")
        n_total += len(info)
    assert n_total >= 3
    det.close(); eng.close()


def test_lane_detector_dropin():
    path, W, g = netutil.model("ufldv2_res18")
    ld = D.UltrafastLaneDetectorV2(path, D.LaneModelType.UFLDV2_CULANE, precision="fp32")
    eng = CE.OnnxEngine(path, precision="fp32")
    cfg = ufld_decode.ModelConfig("culane")
    for f in frames(2, 720, 1280, 9):
        ld.DetectFrame(f)
        outs = eng.engine_inference(preprocess.ufld_prepare_input(f, (320, 1600), 0.6))
        wl, ws = ufld_decode.process_output(outs, cfg, 1280, 720)
        pc.check_lanes(list(ld.lane_info.lanes_points), ld.lane_info.lanes_status, wl, ws, tol_px=1)
        st, area = ufld_decode.lanes_area(wl, ws, 720, True)
        assert ld.lane_info.area_status == st
    ld.close(); eng.close()


def test_lane_detector_device_geometry():
    """enable_device_geometry: area polygon / bird-view points / curvature from the device equal the host route
    (LaneInfo from the decoder + PerspectiveTransformation on the host), frame by frame."""
    A = importlib.import_module("adas_amd.analysis")
    path, W, g = netutil.model("ufldv2_res18")
    ld_host = D.UltrafastLaneDetectorV2(path, D.LaneModelType.UFLDV2_CULANE, precision="fp32")
    ld_dev = D.UltrafastLaneDetectorV2(path, D.LaneModelType.UFLDV2_CULANE, precision="fp32")
    tv = A.PerspectiveTransformation((1280, 720))
    ld_dev.enable_device_geometry(tv)
    for f in frames(2, 720, 1280, 21):
        ld_host.DetectFrame(f)
        ld_dev.DetectFrame(f)
        assert ld_dev.lane_info.area_status == ld_host.lane_info.area_status
        a, b = np.asarray(ld_dev.lane_info.area_points, np.int64).reshape(-1, 2), np.asarray(ld_host.lane_info.area_points, np.int64).reshape(-1, 2)
        assert a.shape == b.shape and np.abs(a - b).max(initial=0) <= 1
        bird = [np.asarray(tv.transformToBirdViewPoints(l), np.int64).reshape(-1, 2) for l in ld_host.lane_info.lanes_points]
        for i in range(4):
            np.testing.assert_array_equal(np.asarray(ld_dev.birdview_lanes_points[i], np.int64).reshape(-1, 2), bird[i])
        (d, c), off = tv.calcCurveAndOffset((720, 1280), bird[1], bird[2])
        (d2, c2), off2 = ld_dev.curve_and_offset
        assert d2 == d
        if d is not None:
            assert c2 == pytest.approx(c, rel=1e-7) and off2 == pytest.approx(off, rel=1e-7, abs=1e-9)
    ld_host.close(); ld_dev.close()


def test_lane_detector_v1_dropin():
    """UltrafastLaneDetector (UFLD v1): device pre-processing (plain resize, crop_ratio 1) + network + v1 decode vs the
    oracle chain on the same engine's logits; two source sizes so w_ratio/h_ratio (ultrafastLaneDetector.py:80) change."""
    path, W, g = netutil.model("ufld_v1_res18")
    ld = D.UltrafastLaneDetector(path, D.LaneModelType.UFLD_TUSIMPLE, precision="fp32")
    eng = CE.OnnxEngine(path, precision="fp32")
    cfg = ufld_decode.ModelConfigV1("tusimple")
    n_pts = 0
    for (h, w) in ((720, 1280), (1080, 1920)):
        for f in frames(1, h, w, 13):
            ld.DetectFrame(f)
            out = eng.engine_inference(preprocess.ufld_prepare_input(f, (288, 800), 1.0))
            wl, ws = ufld_decode.process_output_v1(out[0], cfg, 800, 288, w, h)
            pc.check_lanes(list(ld.lane_info.lanes_points), ld.lane_info.lanes_status, wl, ws, tol_px=1)
            st, area = ufld_decode.lanes_area(wl, ws, h, True)
            assert ld.lane_info.area_status == st
            n_pts += sum(len(l) for l in wl)
    assert n_pts > 50
    with pytest.raises(Exception, match="can't use"):
        D.UltrafastLaneDetector(path, D.LaneModelType.UFLDV2_CULANE)
    ld.close(); eng.close()


def test_bytetracker_dropin_with_label_strings():
    """demo.py:273-277 call shape: int xyxy boxes, float confs, label strings, frame."""
    rng = np.random.default_rng(11)
    n = 12
    pos = rng.uniform(100, 900, (n, 2)); vel = rng.normal(0, 6, (n, 2)); size = rng.uniform(40, 160, (n, 2))
    labels = [["car", "truck", "bus"][i % 3] for i in range(n)]
    lab_id = {"car": 0, "truck": 1, "bus": 2}
    trk = D.BYTETracker()
    ora = bytetrack.BYTETracker()
    crop_of = {}
    for fidx in range(40):
        pos += vel + rng.normal(0, 1, (n, 2))
        keep = rng.uniform(size=n) > 0.12
        boxes = np.concatenate([pos, pos + size], 1).astype(np.int64)[keep]
        scores = rng.uniform(0.3, 0.95, n)[keep]
        labs = [l for l, k in zip(labels, keep) if k]
        frame = rng.integers(0, 255, (720, 1280, 3)).astype(np.uint8)
        msgs = trk.update(boxes.tolist(), scores.tolist(), labs, frame)
        want = ora.update(boxes, scores, np.array([lab_id[l] for l in labs]))
        for m in msgs:
            # strack.py:131-143 / byteTracker.py:161-168: a track carries ONE crop, taken from the frame it was activated on at its
            # box of that frame (tlwh truncated to int, clipped to the frame); on that frame the box is the detection's
            assert isinstance(m["crops"], list) and len(m["crops"]) == 1 and m["crops"][0].dtype == np.uint8
            if m["start_frame_number"] == m["curr_frame_number"] == fidx + 1:
                x, y, w, h = (int(np.floor(v + 1e-9)) for v in m["tlwh"])
                ref = frame[max(0, y):min(720, y + h), max(0, x):min(1280, x + w), :]
                crop_of[m["track_id"]] = ref.copy()
            assert np.array_equal(m["crops"][0], crop_of[m["track_id"]]), (fidx, m["track_id"])
        assert [m["track_id"] for m in msgs] == [t["track_id"] for t in want["tracked"]], fidx
        for m, t in zip(msgs, want["tracked"]):
            assert m["is_activated"] == t["is_activated"] and m["state"] == t["state"] and m["score"] == t["score"]
            assert lab_id[m["class_id"]] == t["class_id"]
            np.testing.assert_allclose(m["tlwh"], t["tlwh"], rtol=1e-9, atol=1e-7)
        assert [m["track_id"] for m in trk.lost_stracks] == [t["track_id"] for t in want["lost"]]
    assert len(msgs) >= 8
    trk.reset()
    assert trk.update([[10, 10, 50, 50]], [0.9], ["car"], None)[0]["track_id"] == 1     # per-instance counter restarts
    trk.close()


def test_headless_demo_loop(capsys):
    """tools/demo_headless.py: the reference's demo.py loop (detector -> tracker, lanes + device geometry, distance,
    FCWS/LDWS/LKAS state machine) runs end to end on the drop-in classes."""
    import os, runpy, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = sys.argv
    sys.argv = ["demo_headless.py", "--frames", "6"]
    try:
        mod = runpy.run_path(os.path.join(root, "tools", "demo_headless.py"), run_name="demo_headless")
        assert mod["main"]() == 6
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert out.count("frame ") == 6 and "FCWS" in out and "frames/s" in out


def _bf16_rne(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


@pytest.mark.parametrize("hw", [(720, 1280), (533, 1600)], ids=str)
def test_packed_preprocess_equals_rounded_fp32(hw):
    """adas_preprocess_*_packed = the fp32 tensors of adas_preprocess_* rounded to bf16 (nearest even), as (c0,c1,c2,0) NHWC."""
    import ctypes as C
    f = np.stack(list(frames(2, hw[0], hw[1], 17)))
    dc = L.DeviceBuffer.from_array(f)
    for kind, (H, W) in (("yolo", (640, 640)), ("ufld", (320, 1600))):
        d32 = L.DeviceBuffer(2 * 3 * H * W * 4); d16 = L.DeviceBuffer(2 * H * W * 8)
        if kind == "yolo":
            L.check(L.lib().adas_preprocess_yolo(dc.ptr, 2, hw[0], hw[1], d32.ptr, H, W, 1, None))
            L.check(L.lib().adas_preprocess_yolo_packed(dc.ptr, 2, hw[0], hw[1], d16.ptr, H, W, 1, None))
        else:
            L.check(L.lib().adas_preprocess_ufld(dc.ptr, 2, hw[0], hw[1], d32.ptr, H, W, C.c_double(0.6), None))
            L.check(L.lib().adas_preprocess_ufld_packed(dc.ptr, 2, hw[0], hw[1], d16.ptr, H, W, C.c_double(0.6), None))
        a = d32.download((2, 3, H, W), np.float32)
        b = d16.download((2, H, W, 4), np.uint16)
        np.testing.assert_array_equal(b[..., :3], _bf16_rne(a).transpose(0, 2, 3, 1))
        assert not b[..., 3].any()
        d32.free(); d16.free()
    dc.free()


@pytest.mark.parametrize("name,kind,hw,layer", [("yolov8n", "yolo", (640, 640), "model.1.conv"), ("yolov5n", "yolo", (640, 640), "model.1.conv"),
                                                ("yolov8s", "yolo", (640, 640), "model.0.conv"), ("ufldv2_res18", "ufld", (320, 1600), "model.maxpool")])
def test_engine_packed_input_equals_fp32_input(name, kind, hw, layer):
    """The fused first layer fed the packed bf16 tensor computes exactly what it computes from the fp32 seam tensor
    (compared on the first materialised activation behind it)."""
    import ctypes as C
    path, W, g = netutil.model(name)
    e = CE.HipEngine(path, precision="bf16", max_batch=2)
    assert L.lib().adas_engine_accepts_packed_input(e._h) == 1
    dc = L.DeviceBuffer.from_array(np.stack(list(frames(2, 720, 1280, 23))))
    H, Wd = hw
    d32 = L.DeviceBuffer(2 * 3 * H * Wd * 4); d16 = L.DeviceBuffer(2 * H * Wd * 8)
    if kind == "yolo":
        L.check(L.lib().adas_preprocess_yolo(dc.ptr, 2, 720, 1280, d32.ptr, H, Wd, 1, None))
        L.check(L.lib().adas_preprocess_yolo_packed(dc.ptr, 2, 720, 1280, d16.ptr, H, Wd, 1, None))
    else:
        L.check(L.lib().adas_preprocess_ufld(dc.ptr, 2, 720, 1280, d32.ptr, H, Wd, C.c_double(0.6), None))
        L.check(L.lib().adas_preprocess_ufld_packed(dc.ptr, 2, 720, 1280, d16.ptr, H, Wd, C.c_double(0.6), None))
    e.infer_device(d32.ptr, 2, None)
    a = e.fetch_activation(layer, 2)
    e.infer_device_packed(d16.ptr, 2, None)
    b = e.fetch_activation(layer, 2)
    d = np.abs(a - b)
    print(name, layer, a.shape, "max|diff|", float(d.max()), "differing", int((d > 0).sum()), "of", d.size, "max|a|", float(np.abs(a).max()))
    np.testing.assert_array_equal(a, b)
    e.close(); dc.free(); d32.free(); d16.free()


# ---------------------------------------------------------------------------------------------------------------------
class _StubEffdetEngine:
    """Stands in for an EfficientDet engine (EngineBase surface only): returns the seeded (boxes, ids, confs) of a golden case
    and records the tensor it was handed, so the test can check the device pre-processing too."""

    def __init__(self, case, hw):
        self.case, self.hw = case, hw
        self.engine_dtype = np.float32
        self.framework_type, self.providers = "stub", ["test"]
        self.seen = None

    def get_engine_input_shape(self):
        return [1, 3, self.hw[0], self.hw[1]]

    def get_engine_output_shape(self):
        return [[-1, 4], [-1], [-1]], ["boxes", "ids", "scores"]

    def engine_inference(self, x):
        self.seen = np.array(x)
        return [self.case[1].copy(), self.case[2].copy(), self.case[3].copy()]


@pytest.mark.parametrize("case", synth.effdet_cases(), ids=lambda c: c[0])
def test_efficientdet_detector_dropin(case, tmp_path):
    """EfficientdetDetector (efficientdetDetector.py:18-111): device letterbox + BGR mean/std normalisation bit-exact with the oracle,
    device inverse letterbox + score filter bit-exact with the REFERENCE's own outputs (effdet_post.npz), labels incl. 'unknown'."""
    from oracle import effdet_post
    tag, boxes, ids, confs, src, inp, thr = case
    lab = tmp_path / "labels.txt"
    lab.write_text("\n".join("c%d" % i for i in range(80)))
    eng = _StubEffdetEngine(case, inp)
    det = D.EfficientdetDetector(engine=eng, classes_path=str(lab), box_score=thr)
    frame = frames(1, src[0], src[1], 31)[0]
    det.DetectFrame(frame)
    np.testing.assert_array_equal(eng.seen, effdet_post.prepare_input(frame, inp))
    g = np.load(os.path.join(GOLDEN, "effdet_post.npz"))
    info = det.object_info
    assert len(info) == len(g[tag + "_conf"])
    for r, xywh, conf, label, xyxy in zip(info, g[tag + "_xywh"], g[tag + "_conf"], g[tag + "_label"], g[tag + "_xyxy_int"]):
        assert (r.x, r.y, r.width, r.height) == tuple(xywh) and r.conf == conf and r.label == str(label)
        assert r.tolist() == list(xyxy)
    det.close()
