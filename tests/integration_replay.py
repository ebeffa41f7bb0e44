#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- INTEGRATION.md section A executed on the CPU: the REFERENCE's own detector classes on a swapped `coreEngine`.

    python tests/integration_replay.py <fixture.npz> <out.json>         (run by tests/test_integration_replay.py in a fresh interpreter)

The reference's YoloDetector / UltrafastLaneDetectorV2 (/root/reference, imported UNMODIFIED from where they lie) are constructed the
way demo.py constructs them and fed the fixture's frame.  Their `from coreEngine import TensorRTEngine, OnnxEngine`
(ObjectDetector/yoloDetector.py:12,16; ufldDetector/ultrafastLaneDetectorV2.py:7,13) resolves to a module that exposes the surface this
repo's HipEngine recorded on the GPU (tests/golden/record_dropin_replay.py): same attribute names and values (`framework_type`,
`providers`, `engine_dtype`), same shapes / names from `get_engine_input_shape` / `get_engine_output_shape`, and `engine_inference`
handing back the tensors HipEngine returned for this frame -- after checking that the tensor the reference's own pre-processing
produced IS the one the device computed (digest).  Their `object_info` / `lane_info` go to <out.json>; the test compares them with the
device results recorded beside the tensors.

Environment shims (the reference cannot be imported in this image as it is, SURVEY finding 4 / Appendix C) -- none of them touches a
reference source line:
  numba.jit      identity decorator (the jitted NMS then runs as the NumPy code it is)
  cv2            absent here: `resize` (8-bit INTER_LINEAR), `cvtColor(BGR2RGB)` and `dnn.blobFromImage` are provided from
                 oracle/preprocess.py's restatement of OpenCV's algorithm (what the device kernels are bit-exact against), + the constants
                 the modules read at import
  onnxruntime / tensorrt / pycuda / lap   empty modules (imported at module level by packages on the way; nothing replayed calls them)
  numpy==1.22.1 semantics (requirements.txt:2) where NumPy 2.2 differs on this path:
      * legacy promotion (finding 5): `x - 0.5 * w` on float32 scalars is float64 there; as in tests/golden/make_golden.py the detector
        head is handed over float64-widened (exact), which makes yoloDetector.py:132 run in fp64 as it does in the pinned environment
      * `if (kpss != [])` (ObjectDetector/utils.py:92) on an empty array is False there (DeprecationWarning) and raises in NumPy 2.2:
        the `np` the reference's utils module sees builds arrays whose empty truth value is False, as in 1.22
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("ADAS_REFERENCE", "/root/reference")
for _p in (HERE, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _install_cv2():
    from oracle import preprocess
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR, cv2.COLOR_BGR2RGB = 1, 4
    cv2.FONT_HERSHEY_TRIPLEX, cv2.FONT_HERSHEY_SIMPLEX, cv2.LINE_AA = 4, 0, 16

    def resize(img, dsize, interpolation=1):
        assert interpolation == cv2.INTER_LINEAR and img.dtype == np.uint8
        return preprocess.cv_resize_linear_u8(np.ascontiguousarray(img), dsize)

    def cvtColor(img, code):
        assert code == cv2.COLOR_BGR2RGB
        return np.ascontiguousarray(img[:, :, ::-1])

    def blobFromImage(image, scalefactor=1.0, size=None, mean=None, swapRB=False, crop=False):
        assert size == (image.shape[1], image.shape[0]) and not crop and mean is None      # no resize inside the blob call (yoloDetector.py:99-100)
        img = image[:, :, ::-1] if swapRB else image
        blob = (img.astype(np.float64) * scalefactor).astype(np.float32)                   # float pixels x the double scale factor
        return np.ascontiguousarray(blob.transpose(2, 0, 1)[None])
    cv2.resize, cv2.cvtColor = resize, cvtColor
    cv2.dnn = types.SimpleNamespace(blobFromImage=blobFromImage)
    sys.modules["cv2"] = cv2


def _install_stubs(fixture_path):
    d = tempfile.mkdtemp(prefix="adas_replay_stubs_")
    open(os.path.join(d, "numba.py"), "w").write("def jit(*a, **k):\n    def deco(f):\n        return f\n    return deco\n")
    for m in ("onnxruntime", "tensorrt"):
        open(os.path.join(d, m + ".py"), "w").write("")
    os.makedirs(os.path.join(d, "pycuda"))
    for f in ("__init__.py", "driver.py"):
        open(os.path.join(d, "pycuda", f), "w").write("")
    # (ObjectDetector/__init__.py pulls the tracker package in, which imports `lap` at module level; nothing on this path calls it)
    open(os.path.join(d, "lap.py"), "w").write("def lapjv(*a, **k):\n    raise NotImplementedError('not on the replayed path')\n")
    # the module the reference's detectors import their engines from: INTEGRATION.md section A's shim, bound to the replay engine
    open(os.path.join(d, "coreEngine.py"), "w").write(
        "from integration_replay import ReplayEngine\nOnnxEngine = TensorRTEngine = ReplayEngine\n")
    for p in (REF, d):           # the stub directory in front of the reference tree: ITS coreEngine.py is the one that gets imported
        sys.path.insert(0, p)
    _install_cv2()
    np.float = float
    ReplayEngine.fixture = dict(np.load(fixture_path, allow_pickle=False))


class _Np122Array(np.ndarray):
    def __bool__(self):      # numpy 1.22: the truth value of an EMPTY array is False (DeprecationWarning), not an error
        return False if self.size == 0 else bool(np.asarray(self))


class _Np122(object):
    """`np` as the reference's utils module sees it: numpy, with `array` building arrays of the pinned version's empty-truth rule."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def array(*a, **k):
        return np.array(*a, **k).view(_Np122Array)


def _engine_base():
    import importlib
    if "adas_amd" not in sys.modules:
        sys.modules["adas_amd"] = importlib.import_module("vehicle-cv-adas_amd")
    return importlib.import_module("adas_amd.coreEngine").EngineBase


class ReplayEngine(_engine_base()):
    """This repo's EngineBase (constructor checks, framework_type property) with HipEngine's recorded surface behind it."""
    fixture = None
    calls = []

    def __init__(self, model_path):
        super().__init__(model_path)           # missing file / wrong suffix fail exactly as HipEngine's do
        self.kind = "lane" if "lane" in os.path.basename(model_path) else "det"
        fx, k = self.fixture, self.kind
        self.framework_type = str(fx[k + "_framework_type"])
        self.providers = [str(p) for p in np.atleast_1d(fx[k + "_providers"])]
        self.engine_dtype = np.dtype(str(fx[k + "_engine_dtype"])).type
        self._in = [int(v) for v in fx[k + "_input_shape"]]
        if k == "det":
            self._shapes = [[int(v) for v in s] for s in fx["det_output_shapes"]]
        else:
            self._shapes = [[int(v) for v in s[:n]] for s, n in zip(fx["lane_output_shapes"], fx["lane_output_ndims"])]
        self._names = [str(n) for n in fx[k + "_output_names"]]

    def get_engine_input_shape(self):
        return self._in

    def get_engine_output_shape(self):
        return self._shapes, self._names

    def engine_inference(self, input_tensor):
        fx, k = self.fixture, self.kind
        x = np.ascontiguousarray(input_tensor)
        ok = (x.dtype == self.engine_dtype and list(x.shape) == self._in and
              hashlib.sha256(x.tobytes()).hexdigest() == str(fx[k + "_input_sha256"]))
        ReplayEngine.calls.append({"kind": k, "dtype": x.dtype.name, "shape": list(x.shape), "input_is_the_device_tensor": bool(ok)})
        if k == "det":
            return [fx["det_out0"].astype(np.float64)]          # float64-widened: the pinned environment's promotion (module docstring)
        return [fx["lane_out%d" % i] for i in range(4)]


def main(fixture_path, out_path):
    _install_stubs(fixture_path)
    fx = ReplayEngine.fixture
    work = tempfile.mkdtemp(prefix="adas_replay_models_")
    lab = os.path.join(work, "labels.txt")
    open(lab, "w").write("\n".join(str(s) for s in fx["labels"]))
    det_model, lane_model = os.path.join(work, "det_replay.onnx"), os.path.join(work, "lane_replay.trt")     # both suffix branches
    for p in (det_model, lane_model):
        open(p, "wb").write(b"replayed")
    import ObjectDetector.utils as ref_utils
    ref_utils.np = _Np122()
    from ObjectDetector.yoloDetector import YoloDetector
    from ObjectDetector.utils import ObjectModelType
    from ObjectDetector.core import RectInfo
    from TrafficLaneDetector.ufldDetector.ultrafastLaneDetectorV2 import UltrafastLaneDetectorV2
    from TrafficLaneDetector.ufldDetector.utils import LaneModelType
    import coreEngine
    assert coreEngine.OnnxEngine is ReplayEngine and os.path.dirname(os.path.abspath(coreEngine.__file__)) != os.path.abspath(REF)

    frame = fx["frame"]
    det = YoloDetector(model_path=det_model, model_type=ObjectModelType.YOLOV8, classes_path=lab, box_score=0.4, box_nms_iou=0.45)
    det.DetectFrame(frame)
    info = det.object_info
    lane = UltrafastLaneDetectorV2(lane_model, LaneModelType.UFLDV2_CULANE)
    lane.DetectFrame(frame)
    li = lane.lane_info
    out = {
        "calls": ReplayEngine.calls,
        "det": {"input_shapes": [int(v) for v in det.input_shapes], "input_types": np.dtype(det.input_types).name,
                "output_names": list(det.output_names), "engine_class": type(det.engine).__name__,
                "all_rectinfo": all(isinstance(r, RectInfo) for r in info),
                "xywh": [[float(r.x), float(r.y), float(r.width), float(r.height)] for r in info],
                "conf": [float(r.conf) for r in info], "label": [str(r.label) for r in info],
                "xyxy_int": [[int(v) for v in r.tolist()] for r in info]},
        "lane": {"input_shape": [int(v) for v in lane.input_shape], "output_names": list(lane.output_names),
                 "points": [[[int(p[0]), int(p[1])] for p in pts] for pts in li.lanes_points],
                 "status": [bool(s) for s in li.lanes_status], "area_status": bool(li.area_status),
                 "area_points": (np.asarray(li.area_points, np.int64).reshape(-1, 2).tolist() if li.area_status else [])},
    }
    json.dump(out, open(out_path, "w"))


if __name__ == "__main__":
    import integration_replay as _me      # one module object: the coreEngine shim imports `integration_replay`, not `__main__`
    _me.main(sys.argv[1], sys.argv[2])
