"""CPU: vehicle-cv-adas_amd/analysis.py (SURVEY 8f rows f2 + f3) against traces of the reference's own classes
(tests/golden/analysis.json.gz, made by tests/golden/make_golden_analysis.py under stubs)."""
import gzip, importlib, json, os
import numpy as np
import pytest

from conftest import GOLDEN, load_pkg

load_pkg()
A = importlib.import_module("adas_amd.analysis")
D = importlib.import_module("adas_amd.detectors")
G = json.load(gzip.open(os.path.join(GOLDEN, "analysis.json.gz"), "rt"))


def test_distance_points_and_collision():
    g = G["distance"]
    dm = A.SingleCamDistanceMeasure()
    dm.updateDistance([D.RectInfo(r["x"], r["y"], r["w"], r["h"], r["conf"], r["label"]) for r in g["rects"]])
    assert len(dm.distance_points) == len(g["points"]) > 10
    for got, want in zip(dm.distance_points, g["points"]):
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]        # distance bit-exact (same fp64 expression)
    poly = np.array(g["poly"], np.int64)
    assert dm.calcCollisionPoint(poly) == g["collision"]
    assert dm.calcCollisionPoint(np.array([], dtype=object)) is None and g["collision_empty"] is None
    dm.updateDistance([])
    assert dm.distance_points == [] and dm.calcCollisionPoint(poly) is None


def test_point_in_polygon_cases():
    sq = [[0, 0], [10, 0], [10, 10], [0, 10]]
    assert A.point_in_polygon(sq, (5, 5)) == 1 and A.point_in_polygon(sq, (15, 5)) == -1
    assert A.point_in_polygon(sq, (10, 5)) == 0 and A.point_in_polygon(sq, (0, 0)) == 0        # edge, vertex
    concave = [[0, 0], [10, 0], [10, 10], [5, 4], [0, 10]]
    assert A.point_in_polygon(concave, (5, 8)) == -1 and A.point_in_polygon(concave, (2, 3)) == 1
    assert A.point_in_polygon([], (1, 1)) == -1


def test_perspective_points_curvature_offset():
    g = G["perspective"]
    pt = A.PerspectiveTransformation((1280, 720))
    for st in g["steps"]:
        pt.updateTransformParams(g["left"], g["right"], st["mode"])
        np.testing.assert_array_equal(pt.src, np.float32(st["src"]))
        np.testing.assert_allclose(pt.M, np.array(st["M"]), rtol=1e-12, atol=1e-12)
        bl, br = pt.transformToBirdViewPoints(g["left"]), pt.transformToBirdViewPoints(g["right"])
        np.testing.assert_array_equal(bl, np.array(st["bird_left"]))
        np.testing.assert_array_equal(br, np.array(st["bird_right"]))
        (d, c), off = pt.calcCurveAndOffset((720, 1280, 3), bl, br)
        assert d == st["direction"]
        assert c == pytest.approx(st["curvature"], rel=1e-9) and off == pytest.approx(st["offset"], rel=1e-9, abs=1e-12)
    cv = g["curvy"]
    (d, c), off = A.PerspectiveTransformation((1280, 720)).calcCurveAndOffset(np.zeros((720, 1280, 3), np.uint8), np.array(cv["left"]), np.array(cv["right"]))
    assert d == cv["direction"] and c == pytest.approx(cv["curvature"], rel=1e-9) and off == pytest.approx(cv["offset"], rel=1e-9)
    assert list(pt.transformToBirdViewPoints([])) == [] and pt.calcCurveAndOffset((720, 1280), [], []) == ((None, None), None)
    # homography sanity: the four source corners land on the destination corners
    q = np.concatenate([pt.src.astype(np.float64), np.ones((4, 1))], 1) @ pt.M.T
    np.testing.assert_allclose(q[:, :2] / q[:, 2:3], pt.dst, atol=1e-6)


REPLAY = r"""
import gzip, importlib, json, os, sys
sys.path.insert(0, sys.argv[1])
A = importlib.import_module("vehicle-cv-adas_amd.analysis")
g = json.load(gzip.open(sys.argv[2], "rt"))["state_machine"]
tc = A.TaskConditions()
out = []
for inp in g["inputs"]:
    tc.UpdateCollisionStatus(inp["distance"], inp["area"])
    changed = tc.CheckStatus()
    tc.UpdateOffsetStatus(inp["offset"])
    tc.UpdateRouteStatus(inp["direction"], inp["curvature"])
    out.append(dict(collision=tc.collision_msg.name, offset=tc.offset_msg.name, curvature=tc.curvature_msg.name, toggle=tc.toggle_status,
                    transform=tc.transform_status, osc=list(tc.toggle_oscillator_status), counters=dict(tc.toggle_status_counter), check=bool(changed)))
print(json.dumps(out))
"""


def test_state_machine_trace_matches_reference():
    """Replayed in a child interpreter with PYTHONHASHSEED=0, the seed the golden was made under: the reference's window
    direction (taskConditions.py:262) is `max(set(strings), key=list.count)` with all-zero keys = first element of the set's
    iteration order, which depends on the string-hash seed when a window mixes directions (kept bug-compatible)."""
    import subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, PYTHONHASHSEED="0")
    r = subprocess.run([sys.executable, "-c", REPLAY, ROOT, os.path.join(GOLDEN, "analysis.json.gz")], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    got_all = json.loads(r.stdout)
    g = G["state_machine"]
    seen = set()
    for f, (got, want) in enumerate(zip(got_all, g["trace"])):
        assert want["error"] is None
        for k, v in got.items():
            assert v == want[k], (f, k, v, want[k])
        seen |= {got["collision"], got["offset"], got["curvature"]}
    assert len(got_all) == len(g["trace"]) == 400
    # the drive visits every warning level and both lane-keeping sides
    assert {"WARNING", "PROMPT", "NORMAL", "RIGHT", "LEFT", "CENTER"} <= seen
