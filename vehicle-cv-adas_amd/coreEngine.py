"""Drop-in for the reference's engine seam (coreEngine.py:7-186) on MI355X.

Reference callers do `from coreEngine import TensorRTEngine, OnnxEngine` and pick one by file
suffix (ObjectDetector/yoloDetector.py:12,74-77; ufldDetector/ultrafastLaneDetectorV2.py:7,82-85).
This module exports the same names, all bound to HipEngine, with the same surface:

    Engine(model_path)                     raises Exception("The model path [...] can't not found!") if missing
    .get_engine_input_shape()           -> [1, 3, H, W]
    .get_engine_output_shape()          -> (shapes: list, names: list[str])
    .engine_inference(ndarray NCHW)     -> list[ndarray], each with leading batch dim 1 (or B)
    .framework_type / .providers / .engine_dtype

Added (not in the reference): `.infer_device(d_ptr, batch)` keeps outputs in HBM for the GPU-resident
post-processing; `precision=` and `max_batch=` keyword arguments.

Precision (round 6).  An unmodified caller -- `OnnxEngine(path)` -- gets the EXACT mode: "fp16x3" (every value a (hi, lo) pair of
halves, three f16 MFMAs per product, fp32 accumulate), the mode that reproduces the reference's ONNXRuntime-CPU results (north_star:
1e-3 on activations, bit-exact NMS survivors / track ids); a graph the split layout cannot hold (a 16-bit tensor whose channel count
is not a multiple of 8) falls to "fp32" (f32 MFMA), the other exact mode.  A model whose graph input is float16 (the reference's
`*_fp16.onnx` / `*_fp16.trt`, coreEngine.py:168, demo.py:18-29) runs in "fp16": that model IS half precision in the reference too.
`precision="fp16"` / "bf16" are the throughput modes (half storage + 16-bit MFMA, fp32 accumulate) = the reference's TensorRT-fp16
behaviour, NOT its ONNXRuntime-CPU results: a few threshold / NMS-order / lane arg-max decisions per hundred frames differ.
`ADAS_PRECISION=<mode>` overrides the default for a whole process.
There is no CPU execution path: construction fails if libadas_hip.so or a gfx950 device is missing.
"""
import abc
import ctypes as C
import os

import numpy as np

try:
    from . import _lib as L
except ImportError:  # imported as a top-level module named `coreEngine` (package dir on sys.path)
    import _lib as L

MODEL_SUFFIXES = ('.onnx', '.trt', '.hipm')
DEFAULT_PRECISION = os.environ.get("ADAS_PRECISION", "exact")   # "exact": fp16x3, else fp32; float16-I/O models: fp16 (module docstring)


def _container_io_half(path):
    """Bit 16 of the ADASHIP1 header's in_cpad word (csrc/engine.h FileHeader): the source model's graph I/O is float16."""
    try:
        with open(path, "rb") as f:
            hd = f.read(40)       # magic[8], version, n_bufs, n_ops, n_outputs, in_c, in_h, in_w, in_cpad (uint32 each)
        return len(hd) == 40 and hd[:8] == b"ADASHIP1" and bool((int.from_bytes(hd[36:40], "little") >> 16) & 1)
    except OSError:
        return False


class EngineBase(abc.ABC):
    '''
    Same contract as the reference EngineBase (coreEngine.py:7-39); `.hipm` is accepted next to .onnx/.trt.
    '''

    def __init__(self, model_path):
        if not os.path.isfile(model_path):
            raise Exception("The model path [%s] can't not found!" % model_path)
        assert model_path.endswith(MODEL_SUFFIXES), 'Onnx/TensorRT/Hip Parameters must be a .onnx/.trt/.hipm file.'
        self._framework_type = None

    @property
    def framework_type(self):
        if (self._framework_type == None):
            raise Exception("Framework type can't be None")
        return self._framework_type

    @framework_type.setter
    def framework_type(self, value):
        if (not isinstance(value, str)):
            raise Exception("Framework type need be str")
        self._framework_type = value

    @abc.abstractmethod
    def get_engine_input_shape(self):
        return NotImplemented

    @abc.abstractmethod
    def get_engine_output_shape(self):
        return NotImplemented

    @abc.abstractmethod
    def engine_inference(self):
        return NotImplemented


class HipEngine(EngineBase):
    def __init__(self, model_path, precision=None, max_batch=1):
        EngineBase.__init__(self, model_path)
        precision = precision or DEFAULT_PRECISION
        if precision != "exact" and precision not in L.PRECISIONS:
            raise Exception("precision must be one of %s or 'exact', got %r" % (sorted(L.PRECISIONS), precision))
        model_path = self._resolve_container(model_path)
        h = C.c_void_p()
        if precision == "exact":
            # the parity policy of the module docstring: a half model stays half (it is half in the reference), everything else runs
            # in the split precision, or -- when the G8 layout cannot hold the graph -- on the f32 MFMA
            precision = "fp16" if _container_io_half(model_path) else "fp16x3"
            if precision == "fp16x3":
                try:
                    L.check(L.lib().adas_engine_create(os.fsencode(model_path), L.PRECISIONS[precision], int(max_batch), C.byref(h)))
                except L.AdasError as ex:
                    if "needs multiples of 8" not in str(ex):
                        raise
                    precision, h = "fp32", C.c_void_p()
        if not h.value:
            L.check(L.lib().adas_engine_create(os.fsencode(model_path), L.PRECISIONS[precision], int(max_batch), C.byref(h)))
        self._h = h.value
        self.precision, self.max_batch = precision, int(max_batch)
        self.providers = ['HIPExecutionProvider(gfx950)']
        # coreEngine.py:168: a model whose graph input is float16 makes the callers cast their tensor to float16 and hands float16
        # arrays back; every other model is a float32 seam (the 16-bit compute types are internal)
        self.engine_dtype = np.float16 if L.lib().adas_engine_model_io_half(self._h) else np.float32
        self.framework_type = "hip"
        self.__load_engine_interface()

    @staticmethod
    def _resolve_container(model_path):
        """An actual ONNX file (what the reference passes OnnxEngine, coreEngine.py:161-170) is converted once to the
        `.hipm` container next to it (or in the temp dir) by onnx_import; an ADASHIP1 container is used as is."""
        with open(model_path, "rb") as f:
            magic = f.read(8)
        if magic == b"ADASHIP1" or not model_path.endswith(".onnx"):
            return model_path
        try:
            from . import onnx_import
        except ImportError:
            import onnx_import
        st = os.stat(model_path)
        # the tag names the source file state AND the importer/container revision, so a cache written by an older importer is
        # never picked up by a newer one
        tag = "%s.%d.%d.v%d.hipm" % (os.path.basename(model_path), st.st_size, int(st.st_mtime), onnx_import.IMPORTER_VERSION)
        import tempfile
        for d in (os.path.dirname(os.path.abspath(model_path)), tempfile.gettempdir()):
            cached = os.path.join(d, "." + tag)
            if os.path.isfile(cached):
                return cached
            if os.access(d, os.W_OK):
                # one process per GPU loads the same model at the same time (torchrun): convert into a private temp file and
                # publish it with an atomic rename, so no rank ever opens a half-written container
                fd, tmp = tempfile.mkstemp(prefix="." + tag + ".", suffix=".part", dir=d)
                os.close(fd)
                try:
                    onnx_import.convert(model_path, tmp)
                    os.replace(tmp, cached)
                except ValueError:
                    return model_path      # not ONNX either: let the library report the format error (ADAS_ERR_FORMAT)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                return cached
        return model_path

    def __load_engine_interface(self):
        lib = L.lib()
        d = (C.c_int64 * 4)()
        L.check(lib.adas_engine_input_shape(self._h, d))
        self.__input_shape = [list(d)]
        self.__output_shapes, self.__output_names = [], []
        for i in range(lib.adas_engine_num_outputs(self._h)):
            nd = C.c_int()
            L.check(lib.adas_engine_output_shape(self._h, i, d, C.byref(nd)))
            self.__output_shapes.append(list(d)[:nd.value])
            self.__output_names.append(lib.adas_engine_output_name(self._h, i).decode())

    # ---- reference surface
    def get_engine_input_shape(self):
        return self.__input_shape[0]

    def get_engine_output_shape(self):
        return self.__output_shapes, self.__output_names

    def engine_inference(self, input_tensor):
        x = np.ascontiguousarray(input_tensor, dtype=np.float32)
        shp = self.__input_shape[0]
        if x.ndim != 4 or list(x.shape[1:]) != shp[1:]:
            raise Exception("input tensor shape %s does not match the engine input %s" % (list(x.shape), shp))
        batch = x.shape[0]
        outs = [np.empty([batch] + s[1:], np.float32) for s in self.__output_shapes]
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        L.check(L.lib().adas_engine_infer_host(self._h, L.ptr(x), batch, ptrs))
        if self.engine_dtype is not np.float32:      # an fp16 model returns fp16 arrays, as ONNXRuntime does
            outs = [o.astype(self.engine_dtype) for o in outs]
        return outs

    # ---- device-resident extensions
    @property
    def handle(self):
        return self._h

    def infer_device(self, d_input_ptr, batch=1, stream=None):
        L.check(L.lib().adas_engine_infer_device(self._h, d_input_ptr, int(batch), stream))

    def infer_device_packed(self, d_input_ptr, batch=1, stream=None):
        """Input = the (c0,c1,c2,0) 16-bit NHWC tensor of adas_preprocess_*_packed_prec in this engine's precision (fused first
        layer only)."""
        L.check(L.lib().adas_engine_infer_device_packed(self._h, d_input_ptr, int(batch), stream))

    def output_device_ptr(self, index):
        return L.lib().adas_engine_output_device(self._h, index)

    def stats(self):
        fl, wb, nl = C.c_double(), C.c_double(), C.c_int()
        L.check(L.lib().adas_engine_stats(self._h, C.byref(fl), C.byref(wb), C.byref(nl)))
        return dict(flops_per_frame=fl.value, weight_bytes=wb.value, num_layers=nl.value)

    def layer_info(self, i):
        name = C.create_string_buffer(64)
        fl, kind = C.c_double(), C.c_int()
        L.check(L.lib().adas_engine_layer_info(self._h, i, name, 64, C.byref(fl), C.byref(kind)))
        return name.value.decode(), fl.value, kind.value

    def layer_kernel(self, i, batch=1):
        name = C.create_string_buffer(96)
        L.check(L.lib().adas_engine_layer_kernel(self._h, i, int(batch), name, 96))
        return name.value.decode()

    # ---- multi-layer launches (csrc/conv_ml.hip): what the engine decided for a batch size, and the last launch's status
    def prepare(self, batch=1):
        L.check(L.lib().adas_engine_prepare(self._h, int(batch)))

    def ml_info(self, batch=1):
        """{'launches', 'layers', 'items'} of the multi-layer launches prepared for `batch` frames (zeros: none / ADAS_NO_ML=1)."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(L.lib().adas_engine_ml_info(self._h, int(batch), C.byref(a), C.byref(b), C.byref(c)))
        return {"launches": a.value, "layers": b.value, "items": c.value}

    def ml_status(self, batch=1):
        """Synchronises; raises when a dependency wait of the last multi-layer launch timed out (every wait is bounded)."""
        w = C.c_uint32()
        L.check(L.lib().adas_engine_ml_status(self._h, int(batch), C.byref(w)))
        return int(w.value)

    def launch_count(self, batch=1):
        return int(L.lib().adas_engine_launch_count(self._h, int(batch)))

    def layer_index(self, name):
        for i in range(self.stats()["num_layers"]):
            if self.layer_info(i)[0] == name:
                return i
        raise KeyError(name)

    def profile(self, d_input_ptr, batch=1, iters=5):
        n = self.stats()["num_layers"]
        ms = np.zeros(n, np.float32)
        nl = C.c_int()
        L.check(L.lib().adas_engine_profile(self._h, d_input_ptr, int(batch), int(iters), L.ptr(ms), n, C.byref(nl)))
        return [(self.layer_info(i) + (float(ms[i]),)) for i in range(n)]

    def fetch_activation(self, layer, batch=1):
        if isinstance(layer, str):
            layer = self.layer_index(layer)
        d = (C.c_int64 * 4)()
        L.check(L.lib().adas_engine_fetch_activation(self._h, layer, batch, None, d))
        out = np.empty(list(d), np.float32)
        L.check(L.lib().adas_engine_fetch_activation(self._h, layer, batch, L.ptr(out), d))
        return out

    def close(self):
        if getattr(self, "_h", None):
            L.lib().adas_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EfficientdetEngine(EngineBase):
    """The exported EfficientDet-D0 graph the reference hands OnnxEngine (efficientdetDetector.py:38): network + in-graph anchor decode and
    NMS, three outputs -- boxes (n, 4) xyxy float32 in input pixels, class ids (n), confidences (n), by descending confidence
    (:68-70).  Here: the "efficientdet-d0" engine graph (models.efficientdet: ten raw head tensors, device-resident) followed by the
    device tail (postproc.EffdetTail).  engine_inference takes the reference's (1, 3, H, W) tensor; a batch returns per-frame lists."""

    OUTPUT_NAMES = ["boxes", "class_ids", "scores"]

    def __init__(self, model_path, precision=None, max_batch=1, score_thr=0.05, iou_thr=0.5, max_det=100, max_candidates=2048):
        EngineBase.__init__(self, model_path)
        try:
            from .postproc import EffdetTail
        except ImportError:
            from postproc import EffdetTail
        self.net = HipEngine(model_path, precision, max_batch)
        shapes, names = self.net.get_engine_output_shape()
        want = [("regression.l%d" % (i // 2)) if i % 2 == 0 else ("classification.l%d" % (i // 2)) for i in range(10)]
        if names != want:
            self.net.close()
            raise Exception("%s is not an EfficientDet head graph (outputs %s)" % (model_path, names))
        self.num_classes = int(shapes[1][2])
        shp = self.net.get_engine_input_shape()
        self.tail = EffdetTail(shp[2:], self.num_classes, score_thr, iou_thr, max_det, max_candidates, max_batch)
        self.max_det, self.max_batch = int(max_det), int(max_batch)
        self.precision = self.net.precision
        self.providers, self.framework_type, self.engine_dtype = self.net.providers, self.net.framework_type, self.net.engine_dtype
        self._x = None

    def get_engine_input_shape(self):
        return self.net.get_engine_input_shape()

    def get_engine_output_shape(self):
        return [[-1, 4], [-1], [-1]], list(self.OUTPUT_NAMES)

    def engine_inference(self, input_tensor):
        x = np.ascontiguousarray(input_tensor, dtype=np.float32)
        shp = self.net.get_engine_input_shape()
        if x.ndim != 4 or list(x.shape[1:]) != shp[1:] or x.shape[0] > self.max_batch:
            raise Exception("input tensor shape %s does not match the engine input %s (max batch %d)" % (list(x.shape), shp, self.max_batch))
        batch = x.shape[0]
        if self._x is None:
            self._x = L.DeviceBuffer(self.max_batch * int(np.prod(shp[1:])) * 4)
        self._x.upload(x)
        self.net.infer_device(self._x.ptr, batch)
        self.tail.run([self.net.output_device_ptr(2 * l) for l in range(5)], [self.net.output_device_ptr(2 * l + 1) for l in range(5)], batch)
        res = [self.tail.fetch(b) for b in range(batch)]
        self.last_candidates = [r["n_candidates"] for r in res]
        if batch == 1:
            r = res[0]
            return [r["boxes"], r["class_id"], r["conf"]]
        return [[r["boxes"] for r in res], [r["class_id"] for r in res], [r["conf"] for r in res]]

    def close(self):
        for k in ("tail", "net"):
            o = getattr(self, k, None)
            if o is not None:
                o.close()
                setattr(self, k, None)
        if getattr(self, "_x", None) is not None:
            self._x.free()
            self._x = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# Names the reference's detector modules import (yoloDetector.py:12,16; ultrafastLaneDetectorV2.py:7,13)
OnnxEngine = HipEngine
TensorRTEngine = HipEngine
