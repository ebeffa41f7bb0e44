"""Thin object wrappers over the post-processing entry points of the C ABI.

YoloPost      <- YoloDetector.__process_output + Scaler.convert_boxes_coordinate + NMS + get_nms_results
                 (ObjectDetector/yoloDetector.py:104-157, utils.py:70-87,105-256)
UfldDecode    <- UltrafastLaneDetectorV2.__process_output (ultrafastLaneDetectorV2.py:114-181)
DeviceTracker <- BYTETracker.update/reset (ObjectTracker/byteTrack/byteTracker.py:62-200)
"""
import ctypes as C

import numpy as np

from . import _lib as L


def letterbox(src_hw, dst_hw, keep_ratio=True):
    """Scaler.process_image geometry (utils.py:42-68) -> dict(pad=(h,w), ratio=(h,w))."""
    p = L.YoloPostParams()
    L.check(L.lib().adas_letterbox_params(int(src_hw[0]), int(src_hw[1]), int(dst_hw[0]), int(dst_hw[1]),
                                          1 if keep_ratio else 0, C.byref(p)))
    return dict(pad=(p.pad_h, p.pad_w), ratio=(p.ratio_h, p.ratio_w))


class YoloPost:
    def __init__(self, layout, num_anchors, num_classes, box_score, iou_thr, lb, nms_mode=L.NMS_REFERENCE,
                 max_candidates=1024, max_batch=1, input_hw=None):
        """input_hw: network input size; required for HEAD_V5_LITE (the grid decode of yoloDetector.py:35-49)."""
        p = L.YoloPostParams(layout, num_anchors, num_classes, nms_mode, box_score, iou_thr, int(lb["pad"][0]),
                             int(lb["pad"][1]), float(lb["ratio"][0]), float(lb["ratio"][1]), max_candidates, 0)
        self.params, self.max_batch, self.cap = p, max_batch, max_candidates
        h = C.c_void_p()
        L.check(L.lib().adas_yolo_post_create(C.byref(p), max_batch, C.byref(h)))
        self.h = h.value
        if input_hw is not None:
            try:
                L.check(L.lib().adas_yolo_post_set_input_size(self.h, int(input_hw[0]), int(input_hw[1])))
            except Exception:
                self.close()
                raise
        self.head_elems = (4 + num_classes) * num_anchors if layout == L.HEAD_V8 else (5 + num_classes) * num_anchors

    def run_device(self, d_head_ptr, batch=1, stream=None):
        L.check(L.lib().adas_yolo_post_run(self.h, d_head_ptr, batch, stream))

    def run_host(self, heads):
        """heads: [batch, ...] fp32 in the reference layout; uploaded, processed on the GPU."""
        heads = np.ascontiguousarray(heads, np.float32)
        batch = heads.shape[0]
        assert heads[0].size == self.head_elems
        buf = L.DeviceBuffer.from_array(heads)
        try:
            self.run_device(buf.ptr, batch)
            return [self.fetch(b) for b in range(batch)]
        finally:
            buf.free()

    def fetch(self, frame=0):
        cap = self.cap
        cnt = L.YoloCounts()
        o = dict(cand_anchor=np.zeros(cap, np.int32), cand_xywh=np.zeros((cap, 4)), cand_conf=np.zeros(cap),
                 cand_cls=np.zeros(cap, np.int32), keep=np.zeros(cap, np.int32), xywh=np.zeros((cap, 4)),
                 conf=np.zeros(cap), class_id=np.zeros(cap, np.int32), xyxy_int=np.zeros((cap, 4), np.int32))
        rc = L.lib().adas_yolo_post_fetch(self.h, frame, C.byref(cnt), *[L.ptr(o[k]) for k in (
            "cand_anchor", "cand_xywh", "cand_conf", "cand_cls", "keep", "xywh", "conf", "class_id", "xyxy_int")])
        n, k = cnt.n_candidates, cnt.n_keep
        res = dict(n_found=cnt.n_found, overflow=bool(cnt.flags & 1))
        for key in ("cand_anchor", "cand_xywh", "cand_conf", "cand_cls"):
            res[key] = o[key][:n]
        for key in ("keep", "xywh", "conf", "class_id", "xyxy_int"):
            res[key] = o[key][:k]
        res["rc"] = rc
        return res

    def fetch_dets(self, frame=0):
        """The survivors only, as one packed device-to-host message (adas_yolo_post_fetch_dets): the keys of fetch() that
        get_nms_results reads (yoloDetector.py:141-157), same values."""
        cap = self.cap
        cnt = L.YoloCounts()
        o = dict(keep=np.zeros(cap, np.int32), xywh=np.zeros((cap, 4)), conf=np.zeros(cap), class_id=np.zeros(cap, np.int32),
                 xyxy_int=np.zeros((cap, 4), np.int32))
        rc = L.lib().adas_yolo_post_fetch_dets(self.h, frame, C.byref(cnt), *[L.ptr(o[k]) for k in ("keep", "xywh", "conf", "class_id", "xyxy_int")])
        res = dict(n_found=cnt.n_found, n_candidates=cnt.n_candidates, overflow=bool(cnt.flags & 1), rc=rc)
        for key, a in o.items():
            res[key] = a[:cnt.n_keep]
        return res

    def device_views(self):
        v = [C.c_void_p() for _ in range(4)]
        L.check(L.lib().adas_yolo_post_device_views(self.h, *[C.byref(x) for x in v]))
        return dict(xyxy=v[0].value, score=v[1].value, cls=v[2].value, counts=v[3].value)

    def close(self):
        if getattr(self, "h", None):
            L.lib().adas_yolo_post_destroy(self.h)
            self.h = None

    __del__ = close


class EffdetTail:
    """The in-graph tail of the exported EfficientDet-D0 (adas_effdet_tail_*): anchor decode + score threshold + per-class NMS over the ten
    raw head tensors of the "efficientdet-d0" engine graph -> (boxes xyxy float32, class ids, confidences) per frame, the three arrays
    EfficientdetDetector.__process_output reads (efficientdetDetector.py:68-70)."""

    def __init__(self, in_hw, num_classes=90, score_thr=0.05, iou_thr=0.5, max_det=100, max_candidates=2048, max_batch=1, anchor_scale=4.0):
        p = L.EffdetTailParams(int(in_hw[0]), int(in_hw[1]), int(num_classes), int(max_candidates), int(max_det), 0, float(score_thr), float(iou_thr),
                               float(anchor_scale))
        self.max_det, self.max_batch = int(max_det), int(max_batch)
        h = C.c_void_p()
        L.check(L.lib().adas_effdet_tail_create(C.byref(p), max_batch, C.byref(h)))
        self.h = h.value

    def run(self, reg_ptrs, cls_ptrs, batch, stream=None):
        """reg_ptrs / cls_ptrs: five device pointers each (pyramid levels 3..7), [batch][cells * 9][4 | num_classes] float32."""
        reg = (C.c_void_p * 5)(*[int(p) for p in reg_ptrs])
        cls = (C.c_void_p * 5)(*[int(p) for p in cls_ptrs])
        L.check(L.lib().adas_effdet_tail_run(self.h, reg, cls, int(batch), stream))

    def fetch(self, frame=0):
        n, nc = C.c_int32(), C.c_int32()
        boxes = np.zeros((self.max_det, 4), np.float32); ids = np.zeros(self.max_det, np.int32); conf = np.zeros(self.max_det, np.float32)
        L.check(L.lib().adas_effdet_tail_fetch(self.h, frame, C.byref(n), L.ptr(boxes), L.ptr(ids), L.ptr(conf), C.byref(nc)))
        k = n.value
        return dict(boxes=boxes[:k].copy(), class_id=ids[:k].astype(np.int64), conf=conf[:k].copy(), n_candidates=nc.value)

    def close(self):
        if getattr(self, "h", None):
            L.lib().adas_effdet_tail_destroy(self.h)
            self.h = None

    __del__ = close


class EffdetPost:
    """EfficientdetDetector.__process_output on the device (adas_effdet_post_*): inverse letterbox in float32 + `conf < box_score`
    filter over the exported graph's (boxes, ids, confs) outputs."""

    def __init__(self, box_score, lb, max_boxes=256, max_batch=1):
        p = L.EffdetPostParams(float(box_score), int(lb["pad"][0]), int(lb["pad"][1]), float(lb["ratio"][0]), float(lb["ratio"][1]),
                               int(max_boxes), 0)
        self.cap, self.max_batch = int(max_boxes), int(max_batch)
        h = C.c_void_p()
        L.check(L.lib().adas_effdet_post_create(C.byref(p), max_batch, C.byref(h)))
        self.h = h.value

    def run_host(self, frames):
        """frames: list (<= max_batch) of (boxes (n,4), ids (n,), confs (n,)) host arrays -> list of result dicts."""
        B, cap = len(frames), self.cap
        boxes = np.zeros((B, cap, 4), np.float32); ids = np.zeros((B, cap), np.int32); confs = np.zeros((B, cap), np.float32)
        counts = np.zeros(B, np.int32)
        for b, (bx, i_, cf) in enumerate(frames):
            n = len(cf)
            if n > cap:
                raise L.AdasError(-1, "frame %d carries %d detections, max_boxes is %d" % (b, n, cap))
            counts[b] = n
            boxes[b, :n] = np.asarray(bx, np.float32).reshape(-1, 4); ids[b, :n] = np.asarray(i_).astype(np.int32); confs[b, :n] = cf
        bufs = [L.DeviceBuffer.from_array(a) for a in (boxes, ids, confs)]
        try:
            L.check(L.lib().adas_effdet_post_run(self.h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, L.ptr(counts), B, None))
            return [self.fetch(b) for b in range(B)]
        finally:
            for x in bufs:
                x.free()

    def fetch(self, frame=0):
        cap = self.cap
        k = C.c_int32()
        xywh = np.zeros((cap, 4), np.float32); conf = np.zeros(cap, np.float32); cls = np.zeros(cap, np.int32); xi = np.zeros((cap, 4), np.int32)
        L.check(L.lib().adas_effdet_post_fetch(self.h, frame, C.byref(k), L.ptr(xywh), L.ptr(conf), L.ptr(cls), L.ptr(xi)))
        n = k.value
        return dict(xywh=xywh[:n], conf=conf[:n], class_id=cls[:n].astype(np.int64), xyxy_int=xi[:n].astype(np.int64))

    def close(self):
        if getattr(self, "h", None):
            L.lib().adas_effdet_post_destroy(self.h)
            self.h = None

    __del__ = close


class UfldDecode:
    def __init__(self, grid_row, cls_row, grid_col, cls_col, img_w, img_h, row_anchor, col_anchor, local_width=1,
                 max_batch=1, num_lanes=4):
        """row_anchor / col_anchor: at least cls_row / cls_col entries; the first cls_* are used, as the reference indexes
        cfg.*_anchor[k] with the network's own k (ultrafastLaneDetectorV2.py:152,170 -- its CurveLanes ModelConfig carries 81
        column anchors for a 41-anchor head).  num_lanes: the tensors' last dimension (4; 10 for the CurveLanes configs)."""
        ra = np.ascontiguousarray(np.asarray(row_anchor, np.float64)[:cls_row])
        ca = np.ascontiguousarray(np.asarray(col_anchor, np.float64)[:cls_col])
        assert len(ra) == cls_row and len(ca) == cls_col, "anchor arrays shorter than the head's anchor counts"
        p = L.UfldParams(grid_row, cls_row, grid_col, cls_col, img_w, img_h, local_width, int(num_lanes), L.ptr(ra), L.ptr(ca))
        self.dims = (grid_row, cls_row, grid_col, cls_col)
        h = C.c_void_p()
        L.check(L.lib().adas_ufld_decode_create(C.byref(p), max_batch, C.byref(h)))
        self.h = h.value

    def run_device(self, ptrs, strides, batch=1, stream=None):
        L.check(L.lib().adas_ufld_decode_run(self.h, *ptrs, *strides, batch, stream))

    def run_host(self, outs):
        """outs = [loc_row (B,G,K,4), loc_col, exist_row, exist_col] fp32 arrays."""
        outs = [np.ascontiguousarray(o, np.float32) for o in outs]
        batch = outs[0].shape[0]
        bufs = [L.DeviceBuffer.from_array(o) for o in outs]
        try:
            self.run_device([b.ptr for b in bufs], [o[0].size for o in outs], batch)
            return [self.fetch(b) for b in range(batch)]
        finally:
            for b in bufs:
                b.free()

    def fetch(self, frame=0):
        pts = np.zeros((4, L.UFLD_MAX_POINTS, 2), np.int32)
        cnt = np.zeros(4, np.int32)
        det = np.zeros(4, np.int32)
        L.check(L.lib().adas_ufld_decode_fetch(self.h, frame, L.ptr(pts), L.ptr(cnt), L.ptr(det)))
        lanes = [[(int(x), int(y)) for x, y in pts[i, :cnt[i]]] for i in range(4)]
        return lanes, [bool(d) for d in det]

    def upload(self, lanes, detected, frame=0):
        """Place lane points decoded elsewhere into a frame slot (4 lists of (x, y), 4 bools)."""
        pts = np.zeros((4, L.UFLD_MAX_POINTS, 2), np.int32)
        cnt = np.asarray([len(l) for l in lanes], np.int32)
        for i, l in enumerate(lanes):
            if len(l):
                pts[i, :len(l)] = np.asarray(l, np.int32).reshape(-1, 2)
        det = np.asarray([1 if d else 0 for d in detected], np.int32)
        L.check(L.lib().adas_ufld_decode_upload(self.h, frame, L.ptr(pts), L.ptr(cnt), L.ptr(det)))

    def close(self):
        if getattr(self, "h", None):
            L.lib().adas_ufld_decode_destroy(self.h)
            self.h = None

    __del__ = close


class Ufld1Decode(UfldDecode):
    """UFLD (v1) decoder: UltrafastLaneDetector.__process_output (ultrafastLaneDetector.py:96-139) on the device."""

    def __init__(self, griding_num, cls_num_per_lane, cfg_img_w, cfg_img_h, input_w, input_h, src_w, src_h, row_anchor,
                 max_batch=1):
        ra = np.ascontiguousarray(row_anchor, np.float64)
        p = L.Ufld1Params(griding_num, cls_num_per_lane, cfg_img_w, cfg_img_h, input_w, input_h, src_w, src_h, L.ptr(ra))
        self.dims = (griding_num, cls_num_per_lane)
        h = C.c_void_p()
        L.check(L.lib().adas_ufld1_decode_create(C.byref(p), max_batch, C.byref(h)))
        self.h = h.value

    def set_source_size(self, src_w, src_h):
        L.check(L.lib().adas_ufld1_decode_set_source_size(self.h, int(src_w), int(src_h)))

    def run_device(self, ptr, stride, batch=1, stream=None):
        L.check(L.lib().adas_ufld1_decode_run(self.h, ptr, stride, batch, stream))

    def run_host(self, out):
        """out: (B, G+1, K, 4) fp32."""
        out = np.ascontiguousarray(out, np.float32)
        buf = L.DeviceBuffer.from_array(out)
        try:
            self.run_device(buf.ptr, out[0].size, out.shape[0])
            return [self.fetch(b) for b in range(out.shape[0])]
        finally:
            buf.free()


class LaneGeometry:
    """Ego-lane area polygon, bird-view points, curvature and offset computed on the device from a lane decoder's
    device-resident points (core.py:102-158, perspectiveTransformation.py:120-214)."""
    DIRECTIONS = (None, "L", "R", "F")

    def __init__(self, img_h, bird_wh, M, adjust_lanes=True, max_batch=1):
        p = L.LaneGeometryParams(int(img_h), int(bird_wh[0]), int(bird_wh[1]), 1 if adjust_lanes else 0,
                                 (C.c_double * 9)(*np.asarray(M, np.float64).reshape(9)))
        self.img_h = int(img_h)
        h = C.c_void_p()
        L.check(L.lib().adas_lane_geometry_create(C.byref(p), max_batch, C.byref(h)))
        self.h = h.value

    def set_matrix(self, M):
        m = np.ascontiguousarray(M, np.float64).reshape(9)
        L.check(L.lib().adas_lane_geometry_set_matrix(self.h, L.ptr(m)))

    def run(self, decode, adjust_lanes=True, batch=1, stream=None):
        L.check(L.lib().adas_lane_geometry_run(self.h, decode.h, 1 if adjust_lanes else 0, batch, stream))

    def fetch(self, frame=0):
        res = L.LaneGeometryResult()
        area = np.zeros((2 * self.img_h, 2), np.int32)
        bird = np.zeros((4, L.UFLD_MAX_POINTS, 2), np.int32)
        L.check(L.lib().adas_lane_geometry_fetch(self.h, frame, C.byref(res), L.ptr(area), L.ptr(bird)))
        n = res.n_area_left + res.n_area_right
        return dict(area_status=bool(res.area_status), area_points=area[:n].copy(), n_left=res.n_area_left, n_right=res.n_area_right,
                    bird_points=[bird[i, :res.bird_counts[i]].copy() for i in range(4)],
                    direction=self.DIRECTIONS[res.direction], curvature=res.curvature if res.direction else None,
                    offset=res.offset if res.direction else None)

    def close(self):
        if getattr(self, "h", None):
            L.lib().adas_lane_geometry_destroy(self.h)
            self.h = None

    __del__ = close


class DeviceTracker:
    """n_streams independent ByteTrack instances living on the GPU."""

    def __init__(self, n_streams=1, track_thresh=0.5, track_buffer=30, match_thresh=0.8, frame_rate=30,
                 max_tracks=256, max_dets=256):
        p = L.BytetrackParams(track_thresh, match_thresh, float(frame_rate), track_buffer, max_tracks, max_dets, 0)
        self.n_streams, self.max_tracks = n_streams, max_tracks
        h = C.c_void_p()
        L.check(L.lib().adas_bytetrack_create(C.byref(p), n_streams, C.byref(h)))
        self.h = h.value

    def reset(self, stream=-1):
        L.check(L.lib().adas_bytetrack_reset(self.h, stream))

    def update_host(self, stream, boxes, scores, cls):
        b = np.ascontiguousarray(np.asarray(boxes, np.float64).reshape(-1, 4))
        s = np.ascontiguousarray(np.asarray(scores, np.float64).reshape(-1))
        c = np.ascontiguousarray(np.asarray(cls, np.int32).reshape(-1))
        L.check(L.lib().adas_bytetrack_update_host(self.h, stream, L.ptr(b), L.ptr(s), L.ptr(c), len(s)))

    def update_device(self, views, det_stride, n_streams=None, stream=None, count_stride=4, count_index=2):
        L.check(L.lib().adas_bytetrack_update_device(self.h, views["xyxy"], views["score"], views["cls"],
                                                     views["counts"], det_stride, count_stride, count_index,
                                                     n_streams or self.n_streams, stream))

    def fetch(self, stream=0):
        hdr = L.TrackHeader()
        recs = np.zeros(2 * self.max_tracks, L.TRACK_DTYPE)
        L.check(L.lib().adas_bytetrack_fetch(self.h, stream, C.byref(hdr), L.ptr(recs), len(recs)))
        return hdr, recs[:hdr.n_tracked], recs[hdr.n_tracked:hdr.n_tracked + hdr.n_lost]

    def fetch_trajectories(self, stream=0):
        """STrack.trajectories (strack.py:53,115) of the live tracks in fetch() order (tracked, then lost): a list of (len, 4) tlbr
        arrays, oldest box first."""
        lens = np.zeros(self.max_tracks, np.int32)
        boxes = np.zeros((self.max_tracks, L.TRAJECTORY_LEN, 4), np.float64)
        n = C.c_int32()
        L.check(L.lib().adas_bytetrack_fetch_trajectories(self.h, stream, L.ptr(lens), L.ptr(boxes), self.max_tracks, C.byref(n)))
        return [boxes[k, :lens[k]].copy() for k in range(n.value)]

    def fetch_frame(self, stream, frame):
        """The tracker message of frame `frame` of the last micro-batched step (what BYTETracker.update returned for that frame)."""
        hdr = L.TrackHeader()
        recs = np.zeros(2 * self.max_tracks, L.TRACK_DTYPE)
        L.check(L.lib().adas_bytetrack_fetch_frame(self.h, stream, frame, C.byref(hdr), L.ptr(recs), len(recs)))
        return hdr, recs[:hdr.n_tracked], recs[hdr.n_tracked:hdr.n_tracked + hdr.n_lost]

    def close(self):
        if getattr(self, "h", None):
            L.lib().adas_bytetrack_destroy(self.h)
            self.h = None

    __del__ = close
