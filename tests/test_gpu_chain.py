"""End-to-end parity of the fused device step in the BENCHMARKED precision (fp16) and for the C4 / C5 pipelines, against the
whole fp32 oracle chain (tests/chain_parity.py): u8 camera frames -> adas_pipeline_step_frames (pre-processing, both nets,
decode / NMS, ByteTrack, hipGraph replay on two HIP streams) versus oracle.preprocess -> oracle.nets -> oracle.yolo_post ->
oracle.bytetrack and oracle.ufld_decode.

Contract: yoloDetector.py:126-139 (candidates: best class prob > box_score), :141-157 (fast_soft_nms survivors),
byteTracker.py:62-185 (ids / states), ultrafastLaneDetectorV2.py:115-160 (lane points).

fp32 mode: every discrete decision identical (candidate anchors, survivors, classes, track ids and states).
fp16 mode: a 16-bit network cannot reproduce an fp32 threshold decision on an anchor whose score lies within its rounding error of
the threshold (one half-precision layer alone leaves ~3e-4 relative on a logit; with ~100 candidates out of 8400 anchors a frame
carries on the order of one anchor that close).  What is asserted is therefore how FEW decisions differ, with the measured numbers
printed: candidate anchors differing <= 3 % of the candidates compared, survivor anchors <= 12 %, lane points within 1 px.
"""
import importlib

import numpy as np
import pytest

import netutil
import gpu_api
import chain_parity as CP
from conftest import load_pkg

pytestmark = pytest.mark.gpu
load_pkg()
L = importlib.import_module("adas_amd._lib")
CE = importlib.import_module("adas_amd.coreEngine")
PP = importlib.import_module("adas_amd.postproc")
PL = importlib.import_module("adas_amd.pipeline")
M = importlib.import_module("adas_amd.models")


def _run_chain(tmp_path, det, prec, S, steps, hold, n_sets, use_graph=True, target=100.0, cap=1024, seed=300, emulate=None):
    import bench
    from oracle import preprocess
    pool = [bench.cam_frames(S, seed + i) for i in range(n_sets)]
    seam0 = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for p in pool for f in p])
    det_path, Wd, gd = bench.build_detector(M, CE, det, seam0, str(tmp_path), "chain_%s_%s" % (det, prec), target_per_frame=target, capacity=cap)
    lane_path, Wl, gl = netutil.model("ufldv2_res18")
    pipe = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision=prec, src_hw=(720, 1280), use_graph=use_graph, max_candidates=cap)
    d_pool = [L.DeviceBuffer.from_array(p) for p in pool]
    chain = CP.OracleChain(det, Wd, "ufldv2_res18", Wl)
    st = CP.run_device_chain(pipe, lambda s: PP.YoloPost.fetch(pipe.post, s), lambda s: gpu_api.track_snapshot(*pipe.tracker.fetch(s)),
                             d_pool, pool, chain, steps, hold, list(range(S)))
    pipe.close()
    for b in d_pool:
        b.free()
    out = st.summary()
    if emulate:     # the same schedule through the CPU oracle with storage rounded to `emulate`: the yardstick of _assert_16bit
        ref2 = CP.OracleChain(det, Wd, "ufldv2_res18", Wl)
        ref2._det_cache, ref2._lane_cache = chain._det_cache, chain._lane_cache
        out["emulated"] = CP.run_oracle_vs_oracle(CP.OracleChain(det, Wd, "ufldv2_res18", Wl, emulate=emulate), ref2, pool, steps, hold,
                                                  list(range(S))).summary()
        print("%s storage-rounding emulation (%s) vs fp32 chain: %s" % (det, emulate, {k: out["emulated"][k] for k in (
            "identical_candidate_sets", "identical_survivors", "equivalent_survivor_sets", "identical_track_ids", "equivalent_tracks",
            "lanes_within_1px", "candidate_anchors_differing", "survivor_anchors_differing", "lane_points_off_by_more_than_1px")}))
    print("%s %s S=%d steps=%d graph=%s: %s" % (det, prec, S, steps, use_graph, {k: out[k] for k in (
        "frames", "identical_candidate_sets", "identical_survivor_sets", "identical_survivors", "equivalent_survivor_sets", "identical_track_ids",
        "equivalent_tracks", "track_states_compared", "lanes_within_1px",
        "lane_points_compared", "lane_points_off_by_more_than_1px", "candidates_compared",
        "candidate_anchors_differing", "survivors_compared", "survivor_anchors_differing", "max_conf_diff_on_identical_frames",
        "max_box_diff_px_on_identical_frames", "max_lane_point_diff_px", "first_track_divergence")}))
    if st.mismatch_log:
        print("  mismatches:", st.mismatch_log[:4])
    return out


def _assert_exact(o):
    n = o["frames"]
    assert o["identical_candidate_sets"] == n and o["identical_keep_indices"] == n and o["identical_survivors"] == n, o
    assert o["identical_track_ids"] == o["track_states_compared"] > 0, o
    assert o["lanes_identical_status"] == n and o["lanes_within_1px"] == n, o
    assert o["survivors_compared"] >= 2 * n          # the comparison saw real work
    assert o["max_conf_diff_on_identical_frames"] <= 1e-4 and o["max_box_diff_px_on_identical_frames"] <= 1e-2


def _assert_16bit(o):
    """Bounds on how MANY discrete decisions differ (measured values are printed by _run_chain).  A half-precision network leaves
    ~1e-3 of the anchor-to-anchor logit spread as error (tools/synth_snr.py), so of 8400 anchors with ~100 over the threshold
    about 0.4-1 per frame sits closer to it than that and is decided differently; one such anchor changes the NMS outcome of its
    neighbourhood (one or two survivors).  What the tracker and every consumer see has to hold up regardless: >= 75 % of the compared
    track snapshots EQUIVALENT (same tracks under a consistent id renaming) and >= 90 % of the frames with EQUIVALENT survivor lists
    (one-to-one partners of the same class, IoU >= 0.9, confidence within 2e-2); where the run carries the storage-rounding emulation
    of the same schedule (`emulated`), the device is held to IT as well: no more lost decisions than half storage alone costs.
    The seeded YOLOv8l net does not meet these bounds in fp16 (profiles/r04/layer_drift_yolov8l.txt: flat 3-9e-4 per layer, no kernel
    stands out; its class signal across anchors is ~1 % of the logit magnitude and the calibration stretches the rounding noise with
    it) and gets no lenient variant of this check (round 5): configs[4] is claimed, tested and benchmarked in the modes that reproduce
    the oracle chain exactly (fp32, fp16x3 -- `bench.py --preset c5` defaults to the latter)."""
    n = o["frames"]
    assert o["survivors_compared"] >= 2 * n
    assert o["candidate_anchors_differing"] <= 0.04 * o["candidates_compared"], o
    assert o["survivor_anchors_differing"] <= 4 * n, o
    # the same tracks (state, class, box) under a consistent renaming of ids -- ByteTrack numbers new tracks in detection order, and
    # two detections tied within the 16-bit error swap list places (chain_parity.ChainStats.add_tracks).  Measured on this file's
    # samples: 26 of 32 (YOLOv8n), 12 of 12 (YOLOv8s); the CPU emulation that only rounds storage to half loses as many (below)
    assert o["equivalent_tracks"] >= 0.75 * o["track_states_compared"] > 0, o
    assert o["equivalent_survivor_sets"] >= 0.9 * n, o
    e = o.get("emulated")
    if e is not None:
        # kernel error or 16-bit rounding?  The device may lose no more decisions than the storage-rounding emulation does (+ a
        # small-sample allowance): what fp16 costs here is the format's, not the kernels'
        for k, slack in (("equivalent_tracks", 3), ("equivalent_survivor_sets", 3), ("identical_survivors", 4), ("lanes_within_1px", 4)):
            assert o[k] >= e[k] - slack, (k, o[k], e[k])
        for k, slack in (("candidate_anchors_differing", 6), ("survivor_anchors_differing", 6), ("lane_points_off_by_more_than_1px", 6)):
            assert o[k] <= 2 * e[k] + slack, (k, o[k], e[k])
    assert o["lanes_identical_status"] == n, o
    assert o["lane_points_off_by_more_than_1px"] <= 0.02 * max(1, o["lane_points_compared"]), o
    assert o["max_conf_diff_on_identical_frames"] <= 2e-2 and o["max_box_diff_px_on_identical_frames"] <= 0.5, o


def test_step_frames_fp16_matches_oracle_chain(tmp_path):
    """The north-star pipeline (YOLOv8n + UFLDv2-R18) in the benchmarked precision: 4 streams x 8 steps, three frame sets."""
    o = _run_chain(tmp_path, "yolov8n", "fp16", S=4, steps=8, hold=2, n_sets=3, emulate="fp16")
    _assert_16bit(o)
    assert o["identical_survivor_sets"] >= 0.75 * o["frames"], o


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
def test_step_frames_more_than_two_streams_exact_no_graph(tmp_path, prec):
    """Eager launches (use_graph=False: the section-event path of record_step) and 6 streams: exact against the oracle chain, in the
    fp32 mode and in the split precision (three f16 MFMAs per product, csrc/conv_x3.hip)."""
    o = _run_chain(tmp_path, "yolov8n", prec, S=6, steps=4, hold=2, n_sets=2, use_graph=False)
    _assert_exact(o)


def test_step_frames_split_precision_matches_oracle_chain_exactly(tmp_path):
    """The north-star pipeline in the split precision (fp16x3), hipGraph replay, on the frames the fp16 test above runs: every
    discrete decision of the fp32 oracle chain (candidates, survivors, track ids and states, lane cells) -- the north-star parity
    gate on the 16-bit matrix cores."""
    o = _run_chain(tmp_path, "yolov8n", "fp16x3", S=4, steps=8, hold=2, n_sets=3)
    _assert_exact(o)


@pytest.mark.parametrize("det,prec", [("yolov8s", "fp32"), ("yolov8s", "fp16"), ("yolov8l", "fp32"),
                                      ("yolov8s", "fp16x3"), ("yolov8l", "fp16x3")])
def test_c4_c5_pipelines_match_oracle_chain(tmp_path, det, prec):
    """BASELINE configs[3] / configs[4]: YOLOv8s / YOLOv8l + UFLDv2-R18 + ByteTrack on 1280x720 frames, 2 streams x 6 steps."""
    o = _run_chain(tmp_path, det, prec, S=2, steps=6, hold=2, n_sets=2, seed=410, emulate="fp16" if prec == "fp16" else None)
    if prec in ("fp32", "fp16x3"):
        _assert_exact(o)
    else:
        _assert_16bit(o)


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
def test_micro_batched_step_matches_oracle_chain_exactly(tmp_path, prec):
    """Temporal micro-batching (adas_pipeline_desc.micro_batch): 2 streams x 3 consecutive frames per step through the nets at once,
    tracker updates in temporal order -- exact against the oracle chain consuming the same frames one at a time, the tracker message of
    EVERY frame included (adas_bytetrack_fetch_frame)."""
    import bench
    from oracle import preprocess
    NS, B, steps = 2, 3, 3
    pool = [bench.cam_frames(NS * B, 520 + i) for i in range(2)]
    seam0 = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for p in pool for f in p])
    det_path, Wd, gd = bench.build_detector(M, CE, "yolov8n", seam0, str(tmp_path), "mb", target_per_frame=100.0, capacity=1024)
    lane_path, Wl, gl = netutil.model("ufldv2_res18")
    pipe = PL.AdasPipeline(det_path, lane_path, n_streams=NS, precision=prec, src_hw=(720, 1280), use_graph=True, max_candidates=1024, micro_batch=B)
    d_pool = [L.DeviceBuffer.from_array(p) for p in pool]
    chain = CP.OracleChain("yolov8n", Wd, "ufldv2_res18", Wl)
    st = CP.run_device_chain(pipe, lambda f: PP.YoloPost.fetch(pipe.post, f), lambda s: gpu_api.track_snapshot(*pipe.tracker.fetch(s)),
                             d_pool, pool, chain, steps, 1, list(range(NS)), micro_batch=B, n_streams=NS,
                             fetch_tracks_frame=lambda s, b: gpu_api.track_snapshot(*pipe.tracker.fetch_frame(s, b)))
    pipe.close()
    for b in d_pool:
        b.free()
    o = st.summary()
    print("micro-batch:", o)
    # tracker messages: one per frame (what BYTETracker.update returns every frame, kept by the micro-batched launch) + the live state
    assert o["frames"] == NS * B * steps and o["track_states_compared"] == NS * steps * (B + 1)
    _assert_exact(o)
