"""Pin the oracle (oracle/*.py NumPy restatement) against golden vectors produced
by the reference's own code under stubs (tests/golden/make_golden.py)."""
import gzip, json, os
import numpy as np
import pytest

from conftest import GOLDEN
import synth
from oracle import yolo_post, ufld_decode, bytetrack

MT = {"YOLOV8": "yolov8", "YOLOV5": "yolov5"}


def test_nms_kats():
    g = np.load(os.path.join(GOLDEN, "nms_kat.npz"))
    n = int(g["n_cases"])
    assert n >= 12
    for i in range(n):
        for thr in (45, 30):
            t = f"k{i}_t{thr}"
            keep = yolo_post.fast_soft_nms(g[t + "_boxes"], g[t + "_scores"], thr / 100.0)
            np.testing.assert_array_equal(keep, g[t + "_keep"], err_msg=t)
            alt = yolo_post.fast_nms(g[t + "_boxes"], g[t + "_scores"], thr / 100.0)
            # reference tie order is unspecified (unstable argsort): compare as given unless tied
            sc = g[t + "_scores"]
            if len(np.unique(sc)) == len(sc):
                np.testing.assert_array_equal(alt, g[t + "_keep_alt"], err_msg=t)
            else:
                assert sorted(alt.tolist()) == sorted(g[t + "_keep_alt"].tolist()) or len(alt) == len(g[t + "_keep_alt"])
    # the headline bug-compatibility KATs
    np.testing.assert_array_equal(g["k0_t45_keep"], [1, 2])
    np.testing.assert_array_equal(g["k2_t45_keep"], [1, 2, 2])


@pytest.mark.parametrize("case", synth.yolo_cases(), ids=lambda c: c[0])
def test_yolo_post_chain(case):
    tag, mt, head, lb, bs, iou = case
    g = np.load(os.path.join(GOLDEN, "yolo_post.npz"))
    assert synth.digest(head) == str(g[tag + "_head_sha1"]), "synthetic input drifted from the golden's"
    lbp = yolo_post.letterbox_params(lb["old"], lb["target"])
    assert lbp["new"] == tuple(lb["new"]) and lbp["pad"] == tuple(lb["pad"])
    r = yolo_post.detect_post(head, lbp, MT[mt], bs, iou)
    boxes, cls, conf, _ = yolo_post.process_output(head, MT[mt], bs)
    np.testing.assert_array_equal(boxes, g[tag + "_raw_boxes"])
    np.testing.assert_array_equal(cls, g[tag + "_cls"])
    np.testing.assert_array_equal(conf, g[tag + "_conf"])
    np.testing.assert_array_equal(r["cand_xywh"], g[tag + "_xywh"])
    np.testing.assert_array_equal(r["keep"], g[tag + "_keep"])
    np.testing.assert_array_equal(r["xywh"], g[tag + "_rect_xywh"])
    np.testing.assert_array_equal(r["conf"], g[tag + "_rect_conf"])
    np.testing.assert_array_equal(r["class_id"], g[tag + "_rect_label"])
    np.testing.assert_array_equal(r["xyxy_int"], g[tag + "_rect_xyxy_int"])
    alt = yolo_post.fast_nms(r["cand_xywh"], r["cand_conf"], iou)
    np.testing.assert_array_equal(alt, g[tag + "_keep_alt"])


@pytest.mark.parametrize("case", synth.lite_cases(), ids=lambda c: c[0])
def test_yolo_lite_chain(case):
    """YOLOv5-lite: grid decode (yoloDetector.py:35-49) + the v5 chain, against the reference's own run."""
    tag, head, hw, lb, bs, iou = case
    g = np.load(os.path.join(GOLDEN, "yolo_lite.npz"))
    assert synth.digest(head) == str(g[tag + "_head_sha1"]), "synthetic input drifted from the golden's"
    dec = yolo_post.lite_postprocess(head, hw)
    np.testing.assert_array_equal(dec[::50, :4], g[tag + "_decoded_every50"])
    assert synth.digest(dec[:, :4].copy()) == str(g[tag + "_decoded_sha1"])
    np.testing.assert_array_equal(dec[:, 4:], head[:, 4:])
    lbp = yolo_post.letterbox_params(lb["old"], lb["target"])
    assert lbp["new"] == tuple(lb["new"]) and lbp["pad"] == tuple(lb["pad"])
    r = yolo_post.detect_post(head, lbp, "yolov5_lite", bs, iou, input_hw=hw)
    boxes, cls, conf, _ = yolo_post.process_output(head, "yolov5_lite", bs, hw)
    np.testing.assert_array_equal(boxes, g[tag + "_raw_boxes"])
    np.testing.assert_array_equal(cls, g[tag + "_cls"])
    np.testing.assert_array_equal(conf, g[tag + "_conf"])
    np.testing.assert_array_equal(r["cand_xywh"], g[tag + "_xywh"])
    np.testing.assert_array_equal(r["keep"], g[tag + "_keep"])
    np.testing.assert_array_equal(r["xywh"], g[tag + "_rect_xywh"])
    np.testing.assert_array_equal(r["xyxy_int"], g[tag + "_rect_xyxy_int"])
    np.testing.assert_array_equal(yolo_post.fast_nms(r["cand_xywh"], r["cand_conf"], iou), g[tag + "_keep_alt"])


@pytest.mark.parametrize("case", synth.ufld_cases(), ids=lambda c: c[0])
def test_ufld_decode(case):
    tag, outs, W, H = case
    g = np.load(os.path.join(GOLDEN, "ufld_decode.npz"))
    assert synth.digest(*outs) == str(g[tag + "_in_sha1"])
    cfg = ufld_decode.ModelConfig("culane")
    lanes, status = ufld_decode.process_output(outs, cfg, W, H)
    assert status == g[tag + "_status"].tolist()
    for li in range(4):
        np.testing.assert_array_equal(np.asarray(lanes[li], np.int64).reshape(-1, 2), g[f"{tag}_lane{li}"])
    astat, area = ufld_decode.lanes_area(lanes, status, H, adjust=True)
    assert astat == bool(g[tag + "_area_status"])
    np.testing.assert_array_equal(np.asarray(area, np.int64).reshape(-1, 2), g[tag + "_area"])


@pytest.mark.parametrize("case", synth.curve_cases(), ids=lambda c: c[0])
def test_ufld_decode_curvelanes_geometry(case):
    """10-lane heads of the CurveLanes configuration through the reference's own __process_output (make_golden_curvelanes.py):
    lanes 1,2 / 0,3 decoded, the other six ignored, the 81-entry column-anchor table indexed by the 41-anchor head."""
    tag, outs, W, H = case
    g = np.load(os.path.join(GOLDEN, "ufld_curve_decode.npz"))
    assert "".join(synth.digest(o) for o in outs) == str(g[tag + "_sha1"])
    cfg = ufld_decode.ModelConfig("curvelanes")
    lanes, status = ufld_decode.process_output(outs, cfg, W, H)
    assert status == g[tag + "_status"].tolist()
    for li in range(4):
        np.testing.assert_array_equal(np.asarray(lanes[li], np.int64).reshape(-1, 2), g[f"{tag}_lane{li}"])


@pytest.mark.parametrize("case", synth.effdet_cases(), ids=lambda c: c[0])
def test_effdet_post_chain(case):
    """EfficientDet wrapper restatement vs the reference's own __process_output + Scaler (make_golden_effdet.py)."""
    from oracle import effdet_post
    tag, boxes, ids, confs, src, inp, thr = case
    g = np.load(os.path.join(GOLDEN, "effdet_post.npz"))
    assert synth.digest(boxes, ids, confs) == str(g[tag + "_sha1"])
    r = effdet_post.process_output(boxes, ids, confs, yolo_post.letterbox_params(src, inp), thr)
    np.testing.assert_array_equal(r["xywh"], g[tag + "_xywh"])                # float32, bit-exact
    np.testing.assert_array_equal(r["conf"], g[tag + "_conf"])
    np.testing.assert_array_equal(r["xyxy_int"], g[tag + "_xyxy_int"])
    labels = ["c%d" % i if i < 80 else "unknown" for i in r["class_id"]]
    assert labels == g[tag + "_label"].tolist()


def _load_bt():
    with gzip.open(os.path.join(GOLDEN, "bytetrack.json.gz"), "rt") as f:
        return json.load(f)


@pytest.mark.parametrize("tag", ["t1", "t2", "t3", "t4", "t5", "t6", "t7"])
def test_bytetrack_trace(tag):
    sc = _load_bt()[tag]
    lab = ["car", "person", "truck"]
    trk = bytetrack.BYTETracker()
    for fr, want in zip(sc["frames"], sc["trace"]):
        ids = [lab[i] for i in fr["ids"]] if sc["label_ids"] else fr["ids"]
        got = trk.update(fr["boxes"], fr["scores"], ids)
        assert got["frame_id"] == want["frame_id"] and got["count"] == want["count"]
        for lst in ("tracked", "lost"):
            assert len(got[lst]) == len(want[lst]), (tag, want["frame_id"], lst)
            for a, b in zip(got[lst], want[lst]):
                for k in ("track_id", "state", "is_activated", "class_id", "start_frame", "frame_id", "tracklet_len"):
                    assert a[k] == b[k], (tag, want["frame_id"], lst, k, a, b)
                assert a["score"] == b["score"]
                np.testing.assert_allclose(a["tlwh"], b["tlwh"], rtol=1e-12, atol=1e-12)
    if tag == "t1":  # SURVEY KAT-T1 narrative
        tr = sc["trace"]
        assert [t["track_id"] for t in tr[0]["tracked"]] == [1, 2, 3]
        assert tr[3]["lost"][0]["track_id"] == 3 and tr[4]["count"] == 3
        assert 3 in [t["track_id"] for t in tr[4]["tracked"]]


@pytest.mark.parametrize("tag", ["t1", "t2", "t3", "t4", "t5", "t6", "t7"])
def test_bytetrack_trajectories(tag):
    """STrack.trajectories (strack.py:53,115: the last 30 matched detection boxes, appended by update() only) and filter_trajectories
    (:145-149) of the restatement vs the reference's, at the checkpoints of tests/golden/make_golden_traj.py -- exact fp64."""
    sc = _load_bt()[tag]
    with gzip.open(os.path.join(GOLDEN, "bytetrack_traj.json.gz"), "rt") as f:
        want = json.load(f)[tag]
    lab = ["car", "person", "truck"]
    trk = bytetrack.BYTETracker()
    seen = 0
    for k, fr in enumerate(sc["frames"]):
        ids = [lab[i] for i in fr["ids"]] if sc["label_ids"] else fr["ids"]
        trk.update(fr["boxes"], fr["scores"], ids)
        if str(k) not in want:
            continue
        seen += 1
        tracks = list(trk.tracked_stracks) + list(trk.lost_stracks)
        assert [int(t.track_id) for t in tracks] == [r["track_id"] for r in want[str(k)]]
        for t, r in zip(tracks, want[str(k)]):
            assert len(t.trajectories) == len(r["trajectory"]) <= 30 and (len(t.trajectories) == 30) == r["full"]
            assert [[float(v) for v in b] for b in t.trajectories] == r["trajectory"], (tag, k, r["track_id"])
            kept = t.filter_trajectories((720, 1280), (10, 10))
            assert [i for i, b in enumerate(t.trajectories) if any(b is q for q in kept)] == r["filtered"]
    assert seen == len(want)


@pytest.mark.parametrize("case", synth.ufld1_cases(), ids=lambda c: c[0])
def test_ufld_v1_decode(case):
    """UFLD v1 decoder restatement vs the reference's own __process_output (ultrafastLaneDetector.py:96-139)."""
    tag, cfgname, head, iwh, swh = case
    g = np.load(os.path.join(GOLDEN, "ufld1_decode.npz"))
    assert synth.digest(head) == str(g[tag + "_in_sha1"])
    cfg = ufld_decode.ModelConfigV1(cfgname)
    lanes, status = ufld_decode.process_output_v1(head, cfg, iwh[0], iwh[1], swh[0], swh[1])
    assert status == g[tag + "_status"].tolist()
    for li in range(4):
        np.testing.assert_array_equal(np.asarray(lanes[li], np.int64).reshape(-1, 2), g[f"{tag}_lane{li}"])


# ---------------------------------------------------------------------------------------------------------------------
# Lane networks: oracle/nets.py (BN-folded weights, restated forward) against the reference's OWN parsingNet modules
# (exportLib/ultrafastLaneV2/model_culane.py, exportLib/ultrafastLane/model.py) run un-folded under a torchvision stub
# (tests/golden/make_golden_ufldnet.py).  Pins flatten order, LayerNorm, head slicing / view order and the BN fold.
import ufldnet_params as UP


@pytest.mark.parametrize("case", UP.CASES, ids=lambda c: c[0])
def test_lane_net_oracle_matches_reference_module(case):
    from oracle import nets
    tag, kind, depth, kw = case
    g = np.load(os.path.join(GOLDEN, "ufld_net.npz"))
    if kind == "v2":
        W = UP.fold(UP.ufldv2_state(UP.SEED, depth, **kw))
        x = UP.lane_frame(UP.SEED + 1, kw["in_h"], kw["in_w"])
        taps = {}
        outs = nets.ufldv2_forward(x, W, depth, kw["grid_row"], kw["cls_row"], kw["grid_col"], kw["cls_col"], kw.get("lanes", 4), taps=taps,
                                   fc_norm=kw["fc_norm"])
        pool = taps["fea"].numpy()
    else:
        W = UP.fold(UP.ufld1_state(UP.SEED, depth, **kw))
        x = UP.lane_frame(UP.SEED + 1, 288, 800)
        outs = [nets.ufld_v1_forward(x, W, depth, kw["griding_num"], kw["cls_per_lane"], 4)]
        pool = None
    assert int(g[f"{tag}_n_outputs"]) == len(outs)
    worst = 0.0
    for i, o in enumerate(outs):
        assert tuple(g[f"{tag}_out{i}_shape"]) == tuple(o.shape)
        flat = o.reshape(-1)
        want = g[f"{tag}_out{i}_sample"]
        got = flat[UP.sample_idx(flat.size)]
        scale = float(np.abs(want).max())
        # fp32 on both sides; the BN fold (fp64 -> fp32 weights) and oneDNN's blocking change the rounding, nothing else
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(scale, 1.0), err_msg=f"{tag} out{i}")
        assert abs(float(flat.astype(np.float64).sum()) - float(g[f"{tag}_out{i}_sum"])) <= 2e-4 * float(g[f"{tag}_out{i}_abssum"])
        worst = max(worst, float(np.abs(got - want).max()) / max(scale, 1.0))
    if pool is not None:   # the (C, H, W) flatten of model_culane.py:53 feeds the head in exactly this order
        want = g[f"{tag}_pool_sample"]
        flat = pool.reshape(-1)
        np.testing.assert_allclose(flat[UP.sample_idx(flat.size)], want, rtol=0, atol=2e-4 * max(float(np.abs(want).max()), 1.0))
    print(f"{tag}: max |oracle - reference module| / max|ref| = {worst:.2e}")
