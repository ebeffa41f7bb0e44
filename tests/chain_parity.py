"""End-to-end parity bookkeeping: the fused device step against the whole fp32 oracle chain, frame by frame.

TEST INFRASTRUCTURE (imports oracle/): used by tests/test_gpu_chain.py and by bench.py's `parity.e2e` leg (after the timed region).
The contract (BASELINE.json north_star): "bit-exact for NMS survivor indices / ByteTrack ID assignment" -- what the reference
produces per frame at yoloDetector.py:126-139 (candidates: best class prob > box_score), :141-157 (fast_soft_nms survivors) and
byteTracker.py:62-185 (track ids / states), and the lane points of ultrafastLaneDetectorV2.py:115-160.

The oracle side is CACHED per distinct frame (a held frame is the same input on every step it is shown): the cache holds the
checker's own outputs, never the device's.
"""
import numpy as np

from oracle import nets, preprocess, yolo_post, ufld_decode, bytetrack


class OracleChain:
    """fp32 oracle of one ADAS pipeline: u8 BGR frame -> (detections, lanes); trackers are per stream."""

    def __init__(self, det_name, Wd, lane_name, Wl, src_hw=(720, 1280), det_hw=(640, 640), lane_hw=(320, 1600), crop_ratio=0.6,
                 box_score=0.4, nms_iou=0.45, emulate=None):
        self.det_name, self.Wd, self.lane_name, self.Wl = det_name, Wd, lane_name, Wl
        self.src_hw, self.det_hw, self.lane_hw, self.crop = src_hw, det_hw, lane_hw, crop_ratio
        self.box_score, self.nms_iou = box_score, nms_iou
        self.lb = yolo_post.letterbox_params(src_hw, det_hw)
        self.cfg = ufld_decode.ModelConfig("culane")
        self.trackers = {}                     # stream id -> its own BYTETracker (fresh state)
        self.emulate = emulate          # None: the parity reference.  "fp16"/"bf16": CPU storage-rounding emulation (studies only)
        self._det_cache, self._lane_cache = {}, {}

    def _forward(self, fn, *a, **k):
        nets.EMULATE = self.emulate
        try:
            return fn(*a, **k)
        finally:
            nets.EMULATE = None

    def detections(self, frame, key=None):
        """-> oracle.yolo_post.detect_post() dict of this frame (cached under `key`)."""
        if key is not None and key in self._det_cache:
            return self._det_cache[key]
        x = preprocess.yolo_prepare_input(frame, self.det_hw)
        head = self._forward(nets.detector_forward, self.det_name, x, self.Wd)[0]
        r = yolo_post.detect_post(head, self.lb, nets.head_layout(self.det_name), self.box_score, self.nms_iou)
        if key is not None:
            self._det_cache[key] = r
        return r

    def lanes(self, frame, key=None):
        if self.lane_name is None:
            return None
        if key is not None and key in self._lane_cache:
            return self._lane_cache[key]
        x = preprocess.ufld_prepare_input(frame, self.lane_hw, self.crop)
        outs = self._forward(nets.ufldv2_forward, x, self.Wl, self.lane_name.split("res")[-1])
        r = ufld_decode.process_output(outs, self.cfg, self.src_hw[1], self.src_hw[0])
        if key is not None:
            self._lane_cache[key] = r
        return r

    def track(self, stream, det):
        trk = self.trackers.setdefault(stream, bytetrack.BYTETracker())
        return trk.update(det["xyxy_int"], det["conf"], det["class_id"])


def survivor_anchors(det):
    """Anchor index of every NMS survivor, in keep order (robust to the candidate list's indexing)."""
    keep = np.asarray(det["keep"], np.int64)
    return np.asarray(det["cand_anchor"], np.int64)[keep] if keep.size else np.zeros(0, np.int64)


def track_ids(snap):
    return ([(t["track_id"], t["state"], t["is_activated"], t["class_id"]) for t in snap["tracked"]],
            [(t["track_id"], t["state"], t["class_id"]) for t in snap["lost"]], snap["count"])


def _iou_xywh(a, b):
    """IoU matrix of boxes (x, y, w, h) [top-left + size], plain geometry (a matching criterion, not the reference's "+1" NMS IoU)."""
    a = np.asarray(a, np.float64).reshape(-1, 4); b = np.asarray(b, np.float64).reshape(-1, 4)
    ax2, ay2, bx2, by2 = a[:, 0] + a[:, 2], a[:, 1] + a[:, 3], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
    iw = np.clip(np.minimum(ax2[:, None], bx2[None]) - np.maximum(a[:, 0][:, None], b[:, 0][None]), 0, None)
    ih = np.clip(np.minimum(ay2[:, None], by2[None]) - np.maximum(a[:, 1][:, None], b[:, 1][None]), 0, None)
    inter = iw * ih
    return inter / np.maximum((a[:, 2] * a[:, 3])[:, None] + (b[:, 2] * b[:, 3])[None] - inter, 1e-30)


def unmatched_survivors(got, want, iou_min=0.9, conf_tol=2e-2):
    """Number of survivors on either side without a one-to-one partner of the same class, IoU >= iou_min and |conf diff| <= conf_tol:
    0 means the two detection lists describe the same objects (possibly through different, tied neighbouring anchors)."""
    ng, nw = len(got["keep"]), len(want["keep"])
    if ng == 0 or nw == 0:
        return ng + nw
    iou = _iou_xywh(got["xywh"], want["xywh"])
    ok = (iou >= iou_min) & (np.asarray(got["class_id"])[:, None] == np.asarray(want["class_id"])[None]) & \
         (np.abs(np.asarray(got["conf"])[:, None] - np.asarray(want["conf"])[None]) <= conf_tol)
    used, n_match = set(), 0
    for i in np.argsort(-np.asarray(got["conf"])):
        cand = [j for j in np.argsort(-iou[i]) if ok[i, j] and j not in used]
        if cand:
            used.add(cand[0]); n_match += 1
    return (ng - n_match) + (nw - n_match)


class ChainStats:
    """Counts, over compared frames, how often the device's discrete decisions equal the oracle's."""

    FIELDS = ("frames", "identical_candidate_sets", "identical_keep_indices", "identical_survivor_sets", "identical_survivors",
              "equivalent_survivor_sets",
              "identical_track_ids", "equivalent_tracks", "lanes_identical_status", "lanes_within_1px")

    def __init__(self):
        self.n = dict.fromkeys(self.FIELDS, 0)
        self.n_cand = self.n_cand_sym_diff = self.n_surv = self.n_surv_sym_diff = 0
        self.n_lane_pts = self.n_lane_pts_off = 0
        self.n_surv_unmatched = 0
        self.n_track_checks = 0
        self.max_conf_diff = self.max_box_diff = 0.0
        self.max_lane_px = 0
        self.first_track_divergence = None
        self.idmaps = {}                # stream -> {device track id: oracle track id} (equivalent_tracks)
        self.mismatch_log = []

    def add_detections(self, got, want, ctx=None):
        self.n["frames"] += 1
        ga, wa = np.asarray(got["cand_anchor"], np.int64), np.asarray(want["cand_anchor"], np.int64)
        same_c = ga.shape == wa.shape and np.array_equal(ga, wa) and np.array_equal(got["cand_cls"], want["cand_cls"])
        self.n["identical_candidate_sets"] += int(same_c)
        self.n_cand += len(wa)
        self.n_cand_sym_diff += len(np.setxor1d(ga, wa))
        same_k = np.array_equal(np.asarray(got["keep"], np.int64), np.asarray(want["keep"], np.int64))
        self.n["identical_keep_indices"] += int(same_c and same_k)
        gs, ws = survivor_anchors(got), survivor_anchors(want)
        # the same survivors with the same classes, as a SET (the keep order is the NMS's score order: two survivors whose scores
        # differ by less than the 16-bit error may swap places) ...
        gset = sorted(zip(gs.tolist(), np.asarray(got["class_id"]).tolist()))
        wset = sorted(zip(ws.tolist(), np.asarray(want["class_id"]).tolist()))
        same_set = gset == wset
        self.n["identical_survivor_sets"] += int(same_set)
        # ... and in the same ORDER (what RectInfo consumers and the tracker's id assignment see)
        same_s = gs.shape == ws.shape and np.array_equal(gs, ws) and np.array_equal(got["class_id"], want["class_id"])
        self.n["identical_survivors"] += int(same_s)
        self.n_surv += len(ws)
        self.n_surv_sym_diff += len(np.setxor1d(gs, ws))
        # the same OBJECTS: every survivor has a one-to-one partner of the same class with IoU >= 0.9 and conf within 2e-2 (two
        # neighbouring anchors of one object whose scores tie within the 16-bit error swap roles in the NMS: another anchor id, the
        # same detection)
        um = 0 if same_s else unmatched_survivors(got, want)
        self.n_surv_unmatched += um
        self.n["equivalent_survivor_sets"] += int(um == 0)
        if same_s and len(ws):
            self.max_conf_diff = max(self.max_conf_diff, float(np.abs(np.asarray(got["conf"]) - want["conf"]).max()))
            self.max_box_diff = max(self.max_box_diff, float(np.abs(np.asarray(got["xywh"]) - want["xywh"]).max()))
        if not same_set and len(self.mismatch_log) < 8:
            self.mismatch_log.append({"ctx": ctx, "only_device": np.setdiff1d(gs, ws).tolist()[:6], "only_oracle": np.setdiff1d(ws, gs).tolist()[:6]})
        return same_c, same_s

    def add_tracks(self, got_snap, want_snap, ctx=None):
        same = track_ids(got_snap) == track_ids(want_snap)
        self.n_track_checks += 1
        self.n["identical_track_ids"] += int(same)
        if not same and self.first_track_divergence is None:
            self.first_track_divergence = ctx
        # the same TRACKS under a consistent renaming of ids: ByteTrack numbers new tracks in detection order, so two detections whose
        # confidences tie within the 16-bit error and swap places in the list swap their (future) ids -- every later snapshot of that
        # stream then "differs" although each track follows the same object with the same state.  Tracks are paired by state, class and
        # box (within 2 px); the pairing must be one-to-one and must agree with every earlier snapshot of the stream.
        stream = ctx[1] if ctx is not None and len(ctx) > 1 else 0
        self.n["equivalent_tracks"] += int(same or self._tracks_equivalent(got_snap, want_snap, stream))
        return same

    def _tracks_equivalent(self, got, want, stream):
        idmap = self.idmaps.setdefault(stream, {})
        for key in ("tracked", "lost"):
            g, w = list(got[key]), list(want[key])
            if len(g) != len(w):
                return False
            used = set()
            for t in g:
                best, bd = None, 2.0
                for j, u in enumerate(w):
                    if j in used or u["state"] != t["state"] or u["class_id"] != t["class_id"] or bool(u.get("is_activated", True)) != bool(t.get("is_activated", True)):
                        continue
                    d = float(np.abs(np.asarray(t["tlwh"], np.float64) - np.asarray(u["tlwh"], np.float64)).max())
                    if d <= bd:
                        best, bd = j, d
                if best is None:
                    return False
                used.add(best)
                if idmap.setdefault(t["track_id"], w[best]["track_id"]) != w[best]["track_id"]:
                    return False
        return len(set(idmap.values())) == len(idmap)

    def add_lanes(self, got, want):
        """Per frame: detected flags identical, every lane point within 1 px.  Per point: how many are further off (a row/column
        whose two best grid cells are closer than the 16-bit error picks the other cell: an arg-max decision, tens of pixels)."""
        (gl, gs), (wl, ws) = got, want
        same_status = [bool(s) for s in gs] == [bool(s) for s in ws]
        self.n["lanes_identical_status"] += int(same_status)
        ok = same_status
        if ok:
            for a, b in zip(gl, wl):
                a = np.asarray(a, np.int64).reshape(-1, 2); b = np.asarray(b, np.int64).reshape(-1, 2)
                self.n_lane_pts += len(b)
                if a.shape != b.shape:
                    ok = False
                    self.n_lane_pts_off += abs(len(a) - len(b))
                    continue
                d = np.abs(a - b).max(axis=1) if len(b) else np.zeros(0, np.int64)
                self.max_lane_px = max(self.max_lane_px, int(d.max(initial=0)))
                self.n_lane_pts_off += int((d > 1).sum())
                ok = ok and int(d.max(initial=0)) <= 1
        self.n["lanes_within_1px"] += int(ok)
        return ok

    def summary(self):
        f = max(1, self.n["frames"])
        out = dict(self.n)
        out.update({
            "frac_identical_candidate_sets": round(self.n["identical_candidate_sets"] / f, 4),
            "frac_identical_survivor_sets": round(self.n["identical_survivor_sets"] / f, 4),
            "frac_identical_survivors_in_order": round(self.n["identical_survivors"] / f, 4),
            "frac_equivalent_survivor_sets": round(self.n["equivalent_survivor_sets"] / f, 4),
            "survivors_without_equivalent_partner": self.n_surv_unmatched,
            "track_states_compared": self.n_track_checks,
            "frac_identical_track_ids": round(self.n["identical_track_ids"] / max(1, self.n_track_checks), 4),
            "frac_equivalent_tracks": round(self.n["equivalent_tracks"] / max(1, self.n_track_checks), 4),
            "candidates_compared": self.n_cand, "candidate_anchors_differing": self.n_cand_sym_diff,
            "survivors_compared": self.n_surv, "survivor_anchors_differing": self.n_surv_sym_diff,
            "max_conf_diff_on_identical_frames": float("%.3e" % self.max_conf_diff),
            "max_box_diff_px_on_identical_frames": float("%.3e" % self.max_box_diff),
            "lane_points_compared": self.n_lane_pts, "lane_points_off_by_more_than_1px": self.n_lane_pts_off,
            "max_lane_point_diff_px": self.max_lane_px, "first_track_divergence": self.first_track_divergence,
        })
        return out


def run_device_chain(pipe, fetch_post, fetch_tracks, d_frame_sets, h_frame_sets, chain, steps, hold, streams, src_hw=(720, 1280), crop=0.6,
                     lanes=True, micro_batch=1, n_streams=None, fetch_tracks_frame=None):
    """Drive `pipe` (an AdasPipeline with FRESH tracker state) for `steps` steps over the frame sets (each held `hold` steps) and
    compare streams `streams` with the oracle chain after every step.  fetch_post(f) -> detections dict of frame index f,
    fetch_tracks(s) -> snapshot dict (tests/gpu_api.track_snapshot form).  micro_batch B > 1: a frame set holds B consecutive
    frames of each of the n_streams streams (frame b of stream s at index b * n_streams + s); the oracle tracker of a stream
    consumes them in order and is compared after the last one -- and, with fetch_tracks_frame(s, b) (the per-frame message store of the
    micro-batched tracker launch), after every frame."""
    st = ChainStats()
    B = max(1, int(micro_batch))
    NS = n_streams if n_streams is not None else (len(h_frame_sets[0]) // B)
    for k in range(steps):
        i = (k // hold) % len(d_frame_sets)
        pipe.step_frames(d_frame_sets[i].ptr, src_hw, crop)
        pipe.sync()
        for s in streams:
            want_trk = None
            for b in range(B):
                f = b * NS + s
                frame = h_frame_sets[i][f]
                want = chain.detections(frame, key=(i, f))
                got = fetch_post(f)
                if got.get("overflow"):
                    raise RuntimeError("frame %d step %d: candidate arena overflow in the parity leg" % (f, k))
                st.add_detections(got, want, ctx=[k, s, b])
                want_trk = chain.track(s, want)
                if fetch_tracks_frame is not None:
                    st.add_tracks(fetch_tracks_frame(s, b), want_trk, ctx=[k, s, b])
                if lanes and pipe.decode is not None:
                    st.add_lanes(pipe.decode.fetch(f), chain.lanes(frame, key=(i, f)))
            st.add_tracks(fetch_tracks(s), want_trk, ctx=[k, s])
    return st


def run_oracle_vs_oracle(chain_a, chain_b, h_frame_sets, steps, hold, streams, lanes=True):
    """The same bookkeeping with chain_a in the device's place: how many discrete decisions differ between two ORACLE chains on the frame
    schedule run_device_chain uses (e.g. OracleChain(emulate="fp16") against the fp32 chain: what storage rounding alone costs, the
    yardstick a 16-bit device run is held against)."""
    st = ChainStats()
    for k in range(steps):
        i = (k // hold) % len(h_frame_sets)
        for s in streams:
            frame = h_frame_sets[i][s]
            got, want = chain_a.detections(frame, key=(i, s)), chain_b.detections(frame, key=(i, s))
            st.add_detections(got, want, ctx=[k, s, 0])
            st.add_tracks(chain_a.track(s, got), chain_b.track(s, want), ctx=[k, s])
            if lanes and chain_a.lane_name is not None:
                st.add_lanes(chain_a.lanes(frame, key=(i, s)), chain_b.lanes(frame, key=(i, s)))
    return st
