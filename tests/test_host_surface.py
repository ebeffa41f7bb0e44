"""CPU: the host-side mirror of the reference's public surface (no compute: the HIP library is only loaded and its
symbols checked) and the pre-processing oracle's own invariants."""
import importlib, os, re
import numpy as np
import pytest

from conftest import load_pkg, ROOT
from oracle import preprocess, ufld_decode, yolo_post

load_pkg()
D = importlib.import_module("adas_amd.detectors")
L = importlib.import_module("adas_amd._lib")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "adas_hip.h")).read()
    declared = set(re.findall(r"\b(adas_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"adas_status"}
    lib = L.lib()                       # loads libadas_hip.so, AttributeError on a missing symbol
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(L.exported_symbols()), declared ^ set(L.exported_symbols())


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every struct of include/adas_hip.h compiled by gcc has the size and field offsets of its ctypes mirror in _lib.py
    (an ABI drift between the two would otherwise only show up as garbage on a GPU box)."""
    import ctypes as C, subprocess
    pairs = [("adas_yolo_post_params", L.YoloPostParams), ("adas_yolo_counts", L.YoloCounts), ("adas_ufld_params", L.UfldParams), ("adas_effdet_post_params", L.EffdetPostParams),
             ("adas_effdet_tail_params", L.EffdetTailParams),
             ("adas_ufld1_params", L.Ufld1Params), ("adas_lane_geometry_params", L.LaneGeometryParams),
             ("adas_lane_geometry_result", L.LaneGeometryResult), ("adas_bytetrack_params", L.BytetrackParams),
             ("adas_track_header", L.TrackHeader), ("adas_pipeline_desc", L.PipelineDesc), ("adas_ml_view", L.MlView), ("adas_ml_layer_desc", L.MlLayerDesc)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "adas_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('  printf("adas_track size %zu\\n", sizeof(adas_track));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.rsplit(" ", 1) for l in subprocess.check_output([str(exe)], text=True).strip().splitlines())
    for cname, cls in pairs:
        assert int(got[cname + " size"]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)
    assert int(got["adas_track size"]) == L.TRACK_DTYPE.itemsize


def test_rectinfo_and_defaults_surface():
    r = D.RectInfo(10.7, 20.2, 30.6, 40.9, conf=0.9, label="car")
    assert r.tolist() == [10, 20, 41, 61]                       # int(x + w) evaluated before truncation (core.py:18-23)
    assert r.tolist(format_type="xywh") == [10, 20, 30, 40]
    p = r.pad(2)
    assert (p.x, p.y, p.width, p.height) == (8.7, 18.2, 34.6, 44.9)
    assert D.YoloDetector.get_defaults("box_score") == 0.4 and D.YoloDetector.get_defaults("box_nms_iou") == 0.45
    assert D.YoloDetector.get_defaults("nope").startswith("Unrecognized")
    keep = D.YoloDetector.check_defaults()
    try:
        D.YoloDetector.set_defaults({"model_path": "x"})
        assert D.YoloDetector.check_defaults() == {"model_path": "x"}
    finally:
        D.YoloDetector.set_defaults(keep)
    assert [m.value for m in D.ObjectModelType] == list(range(8)) and D.LaneModelType.UFLDV2_CULANE.value == 3


def test_missing_model_raises_like_the_reference(tmp_path):
    lab = tmp_path / "labels.txt"
    lab.write_text("a\nb\n")
    with pytest.raises(Exception, match="can't not found"):
        D.YoloDetector(model_path=str(tmp_path / "nope.onnx"), model_type=D.ObjectModelType.YOLOV8, classes_path=str(lab))
    with pytest.raises(Exception, match="can't not found"):
        D.UltrafastLaneDetectorV2(str(tmp_path / "nope.trt"), D.LaneModelType.UFLDV2_CULANE)
    with pytest.raises(Exception, match="can't use UFLD_CULANE"):
        D.UltrafastLaneDetectorV2("x.onnx", D.LaneModelType.UFLD_CULANE)


def test_laneinfo_setters_and_config():
    li = D.LaneInfo(np.array([], dtype=object), np.array([], dtype=object), np.array([], dtype=object), False)
    with pytest.raises(Exception):
        li.area_status = True
    with pytest.raises(Exception):
        li.lanes_points = [1, 2]
    with pytest.raises(Exception):
        li.lanes_status = [1, 0]
    li.lanes_status = [True, False]
    c, o = D.ModelConfig(D.LaneModelType.UFLDV2_CULANE), ufld_decode.ModelConfig("culane")
    np.testing.assert_array_equal(c.row_anchor, o.row_anchor)
    np.testing.assert_array_equal(c.col_anchor, o.col_anchor)
    assert c.crop_ratio == 0.6 and c.griding_num == 200


def test_adjust_lanes_points_matches_oracle():
    rng = np.random.default_rng(0)
    ys = np.arange(300, 700, 12)
    left = [(int(500 - 0.4 * (y - 300) + rng.integers(-2, 3)), int(y)) for y in ys]
    right = [(int(700 + 0.5 * (y - 300) + rng.integers(-2, 3)), int(y)) for y in ys]
    gl, gr = D.adjust_lanes_points(left, right, 720)
    wl, wr = ufld_decode.adjust_lanes_points(left, right, 720)
    assert gl == wl and gr == wr and len(gl) > 100
    short = left[:5]
    assert D.adjust_lanes_points(short, right, 720) == (short, right)


def test_preprocess_oracle_invariants():
    rng = np.random.default_rng(1)
    sq = rng.integers(0, 256, (640, 640, 3), dtype=np.uint8)
    x = preprocess.yolo_prepare_input(sq, (640, 640))                       # identity resize: exact
    np.testing.assert_array_equal(x[0], (sq[:, :, ::-1].astype(np.float64) * (1 / 255.0)).astype(np.float32).transpose(2, 0, 1))
    img = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    canvas, new, pad = preprocess.letterbox_image(img, (640, 640))
    lb = yolo_post.letterbox_params((720, 1280), (640, 640))
    assert new == (361, 640) and pad == (139, 0) and tuple(lb["pad"]) == pad
    assert (canvas[:139] == 114).all() and (canvas[139 + 361:] == 114).all()
    const = np.full((300, 500, 3), 77, np.uint8)                            # constant image stays constant under the fixed point
    assert (preprocess.cv_resize_linear_u8(const, (640, 361)) == 77).all()
    up = preprocess.cv_resize_linear_u8(img, (2560, 1440))                  # x2: within the source range, monotone ramps stay ramps
    assert up.shape == (1440, 2560, 3) and up.min() >= img.min() and up.max() <= img.max()
    lane = rng.integers(0, 256, (533, 1600, 3), dtype=np.uint8)            # identity resize, crop bottom 320 rows
    y = preprocess.ufld_prepare_input(lane, (320, 1600), 0.6)
    ref = ((lane[-320:, :, ::-1].astype(np.float32) / 255.0 - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]).astype(np.float32)
    np.testing.assert_array_equal(y[0], ref.transpose(2, 0, 1))


def test_bench_synthetic_camera_frames_are_seeded():
    import bench
    a, b = bench.cam_frames(2, 5, 90, 160), bench.cam_frames(2, 5, 90, 160)
    assert a.shape == (2, 90, 160, 3) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert not np.array_equal(a, bench.cam_frames(2, 6, 90, 160))
