"""ctypes binding of libadas_hip.so -- the only way the Python host side reaches the GPU.

Fails loudly: a missing/unloadable library is an ImportError-like RuntimeError, a missing device is
reported by the library itself (ADAS_ERR_NO_DEVICE); nothing here computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADAS_LIB") or os.path.join(HERE, "libadas_hip.so")   # ADAS_LIB: an instrumented scratch build (tools/)

UFLD_MAX_POINTS = 128
HEAD_V8, HEAD_V5, HEAD_V5_LITE = 0, 1, 2
NMS_REFERENCE, NMS_GREEDY = 0, 1
PREC_BF16, PREC_FP32, PREC_FP16, PREC_FP16X3 = 0, 1, 2, 3
# "fp16x3": split precision -- (hi, lo) half pairs, three f16 MFMAs per product, fp32 accumulate: the fp32 mode's results at speed
PRECISIONS = {"bf16": PREC_BF16, "fp32": PREC_FP32, "fp16": PREC_FP16, "fp16x3": PREC_FP16X3}


class AdasError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libadas_hip error {code}: {msg}")
        self.code = code


class YoloPostParams(C.Structure):
    _fields_ = [("layout", C.c_int32), ("num_anchors", C.c_int32), ("num_classes", C.c_int32), ("nms_mode", C.c_int32),
                ("box_score", C.c_double), ("iou_thr", C.c_double), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
                ("ratio_h", C.c_double), ("ratio_w", C.c_double), ("max_candidates", C.c_int32), ("reserved", C.c_int32)]


class YoloCounts(C.Structure):
    _fields_ = [("n_found", C.c_int32), ("n_candidates", C.c_int32), ("n_keep", C.c_int32), ("flags", C.c_int32)]


class EffdetPostParams(C.Structure):
    _fields_ = [("box_score", C.c_double), ("pad_h", C.c_int32), ("pad_w", C.c_int32), ("ratio_h", C.c_double), ("ratio_w", C.c_double),
                ("max_boxes", C.c_int32), ("reserved", C.c_int32)]


class EffdetTailParams(C.Structure):
    _fields_ = [("in_h", C.c_int32), ("in_w", C.c_int32), ("num_classes", C.c_int32), ("max_candidates", C.c_int32), ("max_det", C.c_int32),
                ("reserved", C.c_int32), ("score_thr", C.c_double), ("iou_thr", C.c_double), ("anchor_scale", C.c_double)]


class UfldParams(C.Structure):
    _fields_ = [("grid_row", C.c_int32), ("cls_row", C.c_int32), ("grid_col", C.c_int32), ("cls_col", C.c_int32),
                ("img_w", C.c_int32), ("img_h", C.c_int32), ("local_width", C.c_int32), ("num_lanes", C.c_int32),
                ("h_row_anchor", C.c_void_p), ("h_col_anchor", C.c_void_p)]


class Ufld1Params(C.Structure):
    _fields_ = [("griding_num", C.c_int32), ("cls_num_per_lane", C.c_int32), ("cfg_img_w", C.c_int32), ("cfg_img_h", C.c_int32),
                ("input_w", C.c_int32), ("input_h", C.c_int32), ("src_w", C.c_int32), ("src_h", C.c_int32),
                ("h_row_anchor", C.c_void_p)]


class LaneGeometryParams(C.Structure):
    _fields_ = [("img_h", C.c_int32), ("bird_w", C.c_int32), ("bird_h", C.c_int32), ("adjust_lanes", C.c_int32), ("M", C.c_double * 9)]


class LaneGeometryResult(C.Structure):
    _fields_ = [("area_status", C.c_int32), ("n_area_left", C.c_int32), ("n_area_right", C.c_int32), ("direction", C.c_int32),
                ("bird_counts", C.c_int32 * 4), ("curvature", C.c_double), ("offset", C.c_double)]


class BytetrackParams(C.Structure):
    _fields_ = [("track_thresh", C.c_double), ("match_thresh", C.c_double), ("frame_rate", C.c_double),
                ("track_buffer", C.c_int32), ("max_tracks", C.c_int32), ("max_dets", C.c_int32), ("reserved", C.c_int32)]


class TrackHeader(C.Structure):
    _fields_ = [("frame_id", C.c_int32), ("id_count", C.c_int32), ("n_tracked", C.c_int32), ("n_lost", C.c_int32),
                ("err", C.c_int32), ("pad", C.c_int32 * 3)]


class PipelineDesc(C.Structure):
    _fields_ = [("detector", C.c_void_p), ("lane", C.c_void_p), ("post", C.c_void_p), ("decode", C.c_void_p),
                ("tracker", C.c_void_p), ("n_streams", C.c_int32), ("use_graph", C.c_int32), ("geometry", C.c_void_p),
                ("micro_batch", C.c_int32), ("reserved", C.c_int32)]


class MlView(C.Structure):
    _fields_ = [("buf", C.c_uint64), ("cs", C.c_int32), ("coff", C.c_int32), ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32)]


class MlLayerDesc(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32), ("res_mode", C.c_int32), ("up_c", C.c_int32), ("halo_bn", C.c_int32),
                ("x", MlView), ("y", MlView), ("res", MlView), ("up", MlView)]


TRACK_DTYPE = np.dtype([("tlwh", "f8", 4), ("score", "f8"), ("track_id", "i4"), ("state", "i4"), ("is_activated", "i4"),
                        ("class_id", "i4"), ("frame_id", "i4"), ("start_frame", "i4"), ("tracklet_len", "i4"), ("traj_len", "i4")])
TRAJECTORY_LEN = 30        # ADAS_TRAJECTORY_LEN

_P = C.c_void_p
_SIGS = {
    "adas_last_error": (C.c_char_p, []),
    "adas_version": (C.c_int, []),
    "adas_device_count": (C.c_int, []),
    "adas_set_device": (C.c_int, [C.c_int]),
    "adas_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "adas_malloc": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "adas_free": (C.c_int, [_P]),
    "adas_memcpy_h2d": (C.c_int, [_P, _P, C.c_size_t]),
    "adas_memcpy_d2h": (C.c_int, [_P, _P, C.c_size_t]),
    "adas_synchronize": (C.c_int, []),
    "adas_host_alloc": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "adas_host_free": (C.c_int, [_P]),
    "adas_timer_create": (C.c_int, [C.POINTER(_P)]),
    "adas_timer_destroy": (C.c_int, [_P]),
    "adas_timer_start": (C.c_int, [_P, _P]),
    "adas_timer_stop": (C.c_int, [_P, _P]),
    "adas_timer_elapsed_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "adas_engine_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "adas_engine_destroy": (C.c_int, [_P]),
    "adas_engine_input_shape": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "adas_engine_num_outputs": (C.c_int, [_P]),
    "adas_engine_output_shape": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "adas_engine_output_name": (C.c_char_p, [_P, C.c_int]),
    "adas_engine_infer_host": (C.c_int, [_P, _P, C.c_int, C.POINTER(_P)]),
    "adas_engine_infer_device": (C.c_int, [_P, _P, C.c_int, _P]),
    "adas_engine_accepts_packed_input": (C.c_int, [_P]),
    "adas_engine_precision": (C.c_int, [_P]),
    "adas_engine_prepare": (C.c_int, [_P, C.c_int]),
    "adas_engine_ml_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "adas_engine_ml_status": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32)]),
    "adas_engine_launch_count": (C.c_int, [_P, C.c_int]),
    "adas_engine_ml_counters": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "adas_debug_ml_plan": (C.c_int, [C.POINTER(MlLayerDesc), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                     C.c_int, C.POINTER(C.c_int32)]),
    "adas_engine_model_io_half": (C.c_int, [_P]),
    "adas_engine_infer_device_packed": (C.c_int, [_P, _P, C.c_int, _P]),
    "adas_engine_output_device": (_P, [_P, C.c_int]),
    "adas_engine_stats": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "adas_engine_profile": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "adas_engine_layer_info": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "adas_engine_layer_kernel": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "adas_engine_fetch_activation": (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(C.c_int64)]),
    "adas_preprocess_yolo": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "adas_preprocess_ufld": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, _P]),
    "adas_preprocess_yolo_packed": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "adas_preprocess_ufld_packed": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, _P]),
    "adas_preprocess_yolo_packed_prec": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "adas_preprocess_ufld_packed_prec": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_double, C.c_int, _P]),
    "adas_letterbox_params": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(YoloPostParams)]),
    "adas_yolo_post_create": (C.c_int, [C.POINTER(YoloPostParams), C.c_int, C.POINTER(_P)]),
    "adas_yolo_post_destroy": (C.c_int, [_P]),
    "adas_yolo_post_set_input_size": (C.c_int, [_P, C.c_int, C.c_int]),
    "adas_yolo_post_run": (C.c_int, [_P, _P, C.c_int, _P]),
    "adas_yolo_post_profile": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "adas_yolo_post_fetch": (C.c_int, [_P, C.c_int, C.POINTER(YoloCounts)] + [_P] * 9),
    "adas_yolo_post_fetch_dets": (C.c_int, [_P, C.c_int, C.POINTER(YoloCounts)] + [_P] * 5),
    "adas_yolo_post_device_views": (C.c_int, [_P] + [C.POINTER(_P)] * 4),
    "adas_yolo_post_scan_views": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "adas_yolo_post_run_prescanned": (C.c_int, [_P, _P, C.c_int, _P]),
    "adas_engine_detect_sink_supported": (C.c_int, [_P]),
    "adas_engine_detect_sink_shape": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "adas_engine_set_detect_sink": (C.c_int, [_P, _P, _P]),
    "adas_pipeline_detect_sink": (C.c_int, [_P]),
    "adas_yolo_post_capacity": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "adas_yolo_post_head_shape": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "adas_effdet_post_create": (C.c_int, [C.POINTER(EffdetPostParams), C.c_int, C.POINTER(_P)]),
    "adas_effdet_post_destroy": (C.c_int, [_P]),
    "adas_effdet_post_run": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P]),
    "adas_effdet_post_fetch": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), _P, _P, _P, _P]),
    "adas_effdet_tail_create": (C.c_int, [C.POINTER(EffdetTailParams), C.c_int, C.POINTER(_P)]),
    "adas_effdet_tail_destroy": (C.c_int, [_P]),
    "adas_effdet_tail_run": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "adas_effdet_tail_fetch": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), _P, _P, _P, C.POINTER(C.c_int32)]),
    "adas_effdet_tail_device_views": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "adas_preprocess_effdet": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "adas_ufld_decode_create": (C.c_int, [C.POINTER(UfldParams), C.c_int, C.POINTER(_P)]),
    "adas_ufld_decode_destroy": (C.c_int, [_P]),
    "adas_ufld_decode_run": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _P]),
    "adas_ufld_decode_fetch": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "adas_ufld_decode_upload": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "adas_ufld1_decode_create": (C.c_int, [C.POINTER(Ufld1Params), C.c_int, C.POINTER(_P)]),
    "adas_ufld1_decode_set_source_size": (C.c_int, [_P, C.c_int, C.c_int]),
    "adas_ufld_decode_kind": (C.c_int, [_P]),
    "adas_ufld_decode_expected_outputs": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "adas_ufld1_decode_run": (C.c_int, [_P, _P, C.c_size_t, C.c_int, _P]),
    "adas_lane_geometry_create": (C.c_int, [C.POINTER(LaneGeometryParams), C.c_int, C.POINTER(_P)]),
    "adas_lane_geometry_destroy": (C.c_int, [_P]),
    "adas_lane_geometry_set_matrix": (C.c_int, [_P, _P]),
    "adas_lane_geometry_run": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "adas_lane_geometry_fetch": (C.c_int, [_P, C.c_int, C.POINTER(LaneGeometryResult), _P, _P]),
    "adas_bytetrack_create": (C.c_int, [C.POINTER(BytetrackParams), C.c_int, C.POINTER(_P)]),
    "adas_bytetrack_destroy": (C.c_int, [_P]),
    "adas_bytetrack_reset": (C.c_int, [_P, C.c_int]),
    "adas_bytetrack_update_host": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int]),
    "adas_bytetrack_update_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "adas_bytetrack_update_device_frames": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "adas_bytetrack_fetch": (C.c_int, [_P, C.c_int, C.POINTER(TrackHeader), _P, C.c_int]),
    "adas_bytetrack_fetch_trajectories": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_int32)]),
    "adas_bytetrack_reserve_frames": (C.c_int, [_P, C.c_int, C.c_int]),
    "adas_bytetrack_fetch_frame": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(TrackHeader), _P, C.c_int]),
    "adas_pipeline_create": (C.c_int, [C.POINTER(PipelineDesc), C.POINTER(_P)]),
    "adas_pipeline_destroy": (C.c_int, [_P]),
    "adas_pipeline_step": (C.c_int, [_P, _P, _P]),
    "adas_pipeline_step_frames": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double]),
    "adas_pipeline_step_frames_host": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double]),
    "adas_pipeline_wait_upload": (C.c_int, [_P]),
    "adas_pipeline_sync": (C.c_int, [_P]),
    "adas_pipeline_timings": (C.c_int, [_P, C.POINTER(C.c_float)]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    """Load libadas_hip.so once and type its entry points.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is not built; run `python vehicle-cv-adas_amd/build.py` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != 0:
        raise AdasError(code, lib().adas_last_error().decode("utf-8", "replace"))
    return code


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A raw HBM allocation owned by Python (adas_malloc / adas_free)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().adas_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_array(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        b.upload(arr)
        return b

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().adas_memcpy_h2d(self.ptr, ptr(arr), arr.nbytes))

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().adas_memcpy_d2h(ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().adas_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class StreamTimer:
    """hipEvents around work launched on `stream` (adas_timer_*): `with StreamTimer() as t: ...launches...; t.ms`."""

    def __init__(self, stream=None):
        self.stream = stream
        h = C.c_void_p()
        check(lib().adas_timer_create(C.byref(h)))
        self.h = h.value

    def __enter__(self):
        check(lib().adas_timer_start(self.h, self.stream))
        return self

    def __exit__(self, *exc):
        check(lib().adas_timer_stop(self.h, self.stream))
        return False

    @property
    def ms(self):
        v = C.c_float()
        check(lib().adas_timer_elapsed_ms(self.h, C.byref(v)))
        return float(v.value)

    def close(self):
        if getattr(self, "h", None):
            lib().adas_timer_destroy(self.h)
            self.h = None

    __del__ = close


class PinnedBuffer:
    """Page-locked host memory (adas_host_alloc) viewed as a NumPy array: frames that adas_pipeline_step_frames_host uploads
    asynchronously must live here for the copy to overlap the previous step."""

    def __init__(self, shape, dtype=np.uint8):
        self.shape, self.dtype = tuple(int(x) for x in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().adas_host_alloc(C.byref(p), self.nbytes))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr)).view(self.dtype).reshape(self.shape)

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            lib().adas_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
