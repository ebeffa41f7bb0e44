"""Fused per-frame ADAS step for S independent video streams on one GPU (C ABI: adas_pipeline_*).

What demo.py:261-281 drives per frame -- detector forward + decode/NMS, tracker update, lane forward +
decode -- runs back to back on one HIP stream (optionally replayed from a hipGraph) with every
intermediate resident in HBM.  Inputs are the engine-seam tensors (NCHW fp32) already on the device.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .coreEngine import HipEngine
from .postproc import YoloPost, UfldDecode, Ufld1Decode, LaneGeometry, DeviceTracker, letterbox
from . import sharding

CULANE = dict(grid_row=200, cls_row=72, grid_col=100, cls_col=81,
              row_anchor=np.linspace(0.42, 1, 72), col_anchor=np.linspace(0, 1, 81))   # ultrafastLaneDetectorV2.py:49-55


class AdasPipeline:
    def __init__(self, det_model=None, lane_model=None, n_streams=1, precision=None, src_hw=(720, 1280),
                 box_score=0.4, nms_iou=0.45, head_layout=L.HEAD_V8, num_classes=None, use_graph=True,
                 max_candidates=512, track=True, lane_cfg=None, nms_mode=L.NMS_REFERENCE, overlap=True, geometry=None, micro_batch=1):
        """geometry: None, or dict(bird_wh=(w, h), M=3x3, adjust_lanes=True) to run the lane-geometry kernel behind the decode.
        micro_batch B > 1: temporal micro-batching (adas_pipeline_desc.micro_batch) -- a step takes B consecutive frames of every
        stream, frame b of stream s at index b * n_streams + s of the input and of every per-frame fetch; the tracker consumes
        them in order.  Throughput mode for few streams per GPU."""
        self.S = n_streams
        self.stream_ids = list(range(n_streams))      # job-wide ids of the local streams (for_rank overrides)
        self.B = max(1, int(micro_batch))
        n_tracks = n_streams
        n_streams = n_streams * self.B          # frames per step through the engines / post / decode handles
        self.det = self.lane = self.post = self.decode = self.tracker = self.geometry = None
        if det_model:
            self.det = HipEngine(det_model, precision, n_streams)
            ishape = self.det.get_engine_input_shape()
            oshape = self.det.get_engine_output_shape()[0][0]
            A = oshape[2] if head_layout == L.HEAD_V8 else oshape[1]
            nc_model = oshape[1] - 4 if head_layout == L.HEAD_V8 else oshape[2] - 5     # yoloDetector.py:110-124
            if num_classes is None:
                num_classes = nc_model
            elif num_classes != nc_model:
                raise ValueError("num_classes=%d but the detector head %s carries %d classes" % (num_classes, oshape, nc_model))
            lb = letterbox(src_hw, ishape[2:])
            self.post = YoloPost(head_layout, A, num_classes, box_score, nms_iou, lb, nms_mode, max_candidates, n_streams)
            if track:
                self.tracker = DeviceTracker(n_tracks, max_dets=max_candidates)
        if lane_model:
            self.lane = HipEngine(lane_model, precision, n_streams)
            cfg = dict(CULANE)
            cfg.update(lane_cfg or {})
            if "griding_num" in cfg:          # UFLD v1 (ultrafastLaneDetector.py ModelConfig): one output tensor
                ish = self.lane.get_engine_input_shape()
                self.decode = Ufld1Decode(cfg["griding_num"], cfg["cls_num_per_lane"], cfg["img_w"], cfg["img_h"], ish[3], ish[2],
                                          src_hw[1], src_hw[0], cfg["row_anchor"], n_streams)
            else:
                nl = self.lane.get_engine_output_shape()[0][0][3]
                self.decode = UfldDecode(cfg["grid_row"], cfg["cls_row"], cfg["grid_col"], cfg["cls_col"], src_hw[1], src_hw[0],
                                         cfg["row_anchor"], cfg["col_anchor"], 1, n_streams, num_lanes=nl)
            if geometry is not None:
                self.geometry = LaneGeometry(src_hw[0], geometry["bird_wh"], geometry["M"], geometry.get("adjust_lanes", True), n_streams)
        d = L.PipelineDesc(self.det.handle if self.det else None, self.lane.handle if self.lane else None,
                           self.post.h if self.post else None, self.decode.h if self.decode else None,
                           self.tracker.h if self.tracker else None, n_tracks, (1 if use_graph else 0) | (0 if overlap else 2),
                           self.geometry.h if self.geometry else None, self.B if self.B > 1 else 0, 0)
        h = C.c_void_p()
        L.check(L.lib().adas_pipeline_create(C.byref(d), C.byref(h)))
        self.h = h.value

    @classmethod
    def for_rank(cls, det_model, lane_model, total_streams, env=None, **kw):
        """The pipeline of ONE rank of a `total_streams`-stream job (one process per GPU): the rank runs the streams
        sharding.streams_of_rank deals it (stream s -> rank s mod world); `.stream_ids[i]` is the job-wide id of local stream i.
        Returns None for a rank that owns no stream (fewer streams than ranks)."""
        env = env or sharding.RankEnv.from_environ()
        ids = sharding.streams_of_rank(int(total_streams), env)
        if not ids:
            return None
        p = cls(det_model, lane_model, n_streams=len(ids), **kw)
        p.stream_ids = ids
        return p

    def step(self, d_det_ptr=None, d_lane_ptr=None):
        L.check(L.lib().adas_pipeline_step(self.h, d_det_ptr, d_lane_ptr))

    def step_frames(self, d_frames_ptr, src_hw, lane_crop_ratio=0.6):
        """One step from n_streams BGR u8 frames (H x W x 3, back to back) in HBM: pre-processing runs inside the step."""
        L.check(L.lib().adas_pipeline_step_frames(self.h, d_frames_ptr, int(src_hw[0]), int(src_hw[1]), float(lane_crop_ratio)))

    def step_frames_host(self, h_frames_ptr, src_hw, lane_crop_ratio=0.6):
        """The same step from HOST frames (pinned: _lib.PinnedBuffer): the upload runs on a copy stream into one of two device
        staging buffers, overlapped with the previous step's compute."""
        L.check(L.lib().adas_pipeline_step_frames_host(self.h, h_frames_ptr, int(src_hw[0]), int(src_hw[1]), float(lane_crop_ratio)))

    def wait_upload(self):
        """Block until the last step_frames_host upload has read its host buffer (then the buffer may be refilled)."""
        L.check(L.lib().adas_pipeline_wait_upload(self.h))

    def sync(self):
        L.check(L.lib().adas_pipeline_sync(self.h))

    def timings(self):
        ms = (C.c_float * 6)()
        L.check(L.lib().adas_pipeline_timings(self.h, ms))
        return dict(zip(("det_net", "det_post", "lane_net", "lane_decode", "tracker", "step"), [float(v) for v in ms]))

    def flops_per_frame(self):
        f = 0.0
        for e in (self.det, self.lane):
            if e:
                f += e.stats()["flops_per_frame"]
        return f

    def close(self):
        if getattr(self, "h", None):
            L.lib().adas_pipeline_destroy(self.h)
            self.h = None
        for o in (self.geometry, self.post, self.decode, self.tracker, self.det, self.lane):
            if o:
                o.close()

    __del__ = close
