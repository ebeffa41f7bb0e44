"""Parity assertions shared by the host-emulation (CPU) and HIP (GPU) suites."""
import numpy as np

TRACK_KEYS = ("track_id", "state", "is_activated", "class_id", "start_frame", "frame_id", "tracklet_len")


def check_yolo(got, want, exact_boxes=True):
    """got: dict from a device/emulated run; want: oracle.yolo_post.detect_post() result."""
    np.testing.assert_array_equal(got["cand_anchor"], want["cand_anchor"])
    np.testing.assert_array_equal(got["cand_cls"], want["cand_cls"])
    np.testing.assert_array_equal(got["cand_conf"], want["cand_conf"])
    np.testing.assert_array_equal(got["cand_xywh"], want["cand_xywh"])          # fp64 bit-exact
    np.testing.assert_array_equal(got["keep"], want["keep"])                    # survivor indices bit-exact
    np.testing.assert_array_equal(got["xywh"], want["xywh"])
    np.testing.assert_array_equal(got["conf"], want["conf"])
    np.testing.assert_array_equal(got["class_id"], want["class_id"])
    np.testing.assert_array_equal(got["xyxy_int"], want["xyxy_int"])


def check_lanes(got_lanes, got_status, want_lanes, want_status, tol_px=0):
    assert list(got_status) == list(want_status)
    n_off = 0
    for g, w in zip(got_lanes, want_lanes):
        g = np.asarray(g, np.int64).reshape(-1, 2); w = np.asarray(w, np.int64).reshape(-1, 2)
        assert g.shape == w.shape
        d = np.abs(g - w)
        assert d.max(initial=0) <= tol_px
        n_off += int((d > 0).sum())
    return n_off


def check_track_frame(got, want, ctx="", rtol=1e-9):
    assert got["frame_id"] == want["frame_id"], ctx
    assert got["count"] == want["count"], (ctx, got["count"], want["count"])
    for lst in ("tracked", "lost"):
        assert [t["track_id"] for t in got[lst]] == [t["track_id"] for t in want[lst]], (ctx, lst)
        for a, b in zip(got[lst], want[lst]):
            for k in TRACK_KEYS:
                assert a[k] == b[k], (ctx, lst, k, a, b)
            assert a["score"] == b["score"], (ctx, lst)
            np.testing.assert_allclose(a["tlwh"], b["tlwh"], rtol=rtol, atol=1e-7, err_msg=str(ctx))
