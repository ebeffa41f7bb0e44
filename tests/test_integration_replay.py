"""INTEGRATION.md section A, executed: the REFERENCE's unmodified YoloDetector / UltrafastLaneDetectorV2 on a swapped `coreEngine`.

CPU test (needs /root/reference: skipped, not failed, where the reference tree does not exist -- e.g. on the GPU box).  The engine the
reference's classes talk to replays what this repo's HipEngine exposed and returned on an MI355X for one 1280x720 frame
(tests/golden/dropin_replay.npz, recorded by tests/golden/record_dropin_replay.py); the reference's results must equal the device
results recorded with it (ObjectDetector/core.py:73-91, ObjectDetector/yoloDetector.py:74-80,159-168,
ufldDetector/ultrafastLaneDetectorV2.py:82-95,183-194).  tests/integration_replay.py holds the replay engine and the environment shims.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = os.environ.get("ADAS_REFERENCE", "/root/reference")
FIXTURE = os.environ.get("ADAS_REPLAY_FIXTURE") or os.path.join(ROOT, "tests", "golden", "dropin_replay.npz")

needs = pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "ObjectDetector")) and os.path.isfile(FIXTURE)),
                           reason="needs the reference tree (/root/reference) and tests/golden/dropin_replay.npz")


@pytest.fixture(scope="module")
def replayed(tmp_path_factory):
    out = tmp_path_factory.mktemp("replay") / "out.json"
    env = dict(os.environ, PYTHONHASHSEED="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "integration_replay.py"), FIXTURE, str(out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out)), dict(np.load(FIXTURE, allow_pickle=False))


@needs
def test_reference_classes_read_the_hip_engine_surface(replayed):
    got, fx = replayed
    # both suffix branches of the reference's dispatch (yoloDetector.py:74-77, ultrafastLaneDetectorV2.py:82-85) landed on the swapped engine,
    # and the tensors their own pre-processing built are bit for bit the ones the device pre-processing fed HipEngine
    assert [c["kind"] for c in got["calls"]] == ["det", "lane"]
    assert all(c["input_is_the_device_tensor"] for c in got["calls"]), got["calls"]
    assert got["det"]["engine_class"] == "ReplayEngine" and got["det"]["input_types"] == str(fx["det_engine_dtype"]) == "float32"
    assert got["det"]["input_shapes"] == [int(v) for v in fx["det_input_shape"]] == [1, 3, 640, 640]       # set_input_details, core.py:73-82
    assert got["det"]["output_names"] == [str(n) for n in fx["det_output_names"]]
    assert got["lane"]["input_shape"] == [int(v) for v in fx["lane_input_shape"]] == [1, 3, 320, 1600]
    assert got["lane"]["output_names"] == [str(n) for n in fx["lane_output_names"]] and len(got["lane"]["output_names"]) == 4   # :93-94
    assert str(fx["det_framework_type"]) == "hip" and "HIPExecutionProvider" in str(np.atleast_1d(fx["det_providers"])[0])


@needs
def test_reference_yolo_detector_on_replayed_engine_equals_device_results(replayed):
    """object_info of the reference's YoloDetector (its own letterbox, decode loop, inverse letterbox, fast_soft_nms, RectInfo) on the
    head HipEngine returned == the device path's object_info for the same frame: every field, exactly."""
    got, fx = replayed
    d = got["det"]
    assert d["all_rectinfo"] and len(d["xywh"]) == len(fx["det_xywh"]) >= 3
    np.testing.assert_array_equal(np.asarray(d["xywh"], np.float64).reshape(-1, 4), fx["det_xywh"])
    np.testing.assert_array_equal(np.asarray(d["conf"], np.float64), fx["det_conf"])
    assert d["label"] == [str(s) for s in fx["det_label"]]
    np.testing.assert_array_equal(np.asarray(d["xyxy_int"], np.int64).reshape(-1, 4), fx["det_xyxy_int"])      # RectInfo.tolist, core.py:18-23


@needs
def test_reference_lane_detector_on_replayed_engine_equals_device_results(replayed):
    """lane_info of the reference's UltrafastLaneDetectorV2 on HipEngine's four outputs == the device path's: same lanes found, same
    number of points per lane, coordinates within 1 px (the 3-tap softmax: NumPy's exp there, expf on the device), same ego-lane
    area decision and polygon (within 2 px: it is a least-squares refit of those points)."""
    got, fx = replayed
    l = got["lane"]
    assert l["status"] == [bool(s) for s in fx["lane_status"]] and sum(l["status"]) >= 1
    for i in range(4):
        want = fx["lane_points%d" % i].reshape(-1, 2)
        have = np.asarray(l["points"][i], np.int64).reshape(-1, 2)
        assert have.shape == want.shape, (i, have.shape, want.shape)
        assert np.abs(have - want).max(initial=0) <= 1, (i, np.abs(have - want).max())
    assert l["area_status"] == bool(fx["lane_area_status"])
    a, b = np.asarray(l["area_points"], np.int64).reshape(-1, 2), fx["lane_area_points"].reshape(-1, 2)
    assert a.shape == b.shape and np.abs(a - b).max(initial=0) <= 2
