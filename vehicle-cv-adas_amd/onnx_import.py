"""ONNX -> `.hipm`: load the model files a reference user already has (first half).

The reference hands `OnnxEngine` an `.onnx` file (coreEngine.py:161-170; exported by ultralytics for the YOLO family,
by TrafficLaneDetector/convertPytorchToONNX.py for UFLDv2).  HipEngine does not interpret ONNX graphs: it runs the
architectures `models.py` builds (hand-written HIP kernels per layer type).  What an ONNX file contributes is the
*weights* and enough shape information to pick the architecture, so this module is

  * a dependency-free reader of the ONNX protobuf wire format (the `onnx` package is not required): initializers
    (float / float16 / double, raw_data or typed fields), Conv / BatchNormalization / Gemm / MatMul nodes in graph order,
    graph input and output shapes;
  * architecture detection (YOLOv8 n/s/m/l/x, YOLOv5 n/s/m/l, UFLDv2 CULane ResNet-18/34) from output shapes, the first
    convolution and initializer names;
  * a weight source for `models.build`: parameters are taken BY NAME when the exporter kept PyTorch names (ultralytics
    exports after Conv+BN fusion keep `model.N.conv.weight`), with BatchNorm folded when BN tensors are present, and BY
    EXECUTION ORDER of the Conv nodes when the exporter's constant folding replaced names by `onnx::Conv_123`
    (torch.onnx.export of an eval-mode ResNet).

Anything that does not match a supported architecture fails loudly with the list of what was found.  PARITY NOTE: no
real checkpoint exists in the reference tree or in this container; the importer is tested on ONNX files written by
tests/onnx_writer.py that follow the two exporters' naming and ordering conventions.
"""
import os
import struct
import sys

import numpy as np

try:
    from . import models as M
except ImportError:  # executed as a script
    import models as M

# Revision of this importer + the container it writes: part of the name of HipEngine's converted-model cache
# (coreEngine.HipEngine._resolve_container), so containers written by an older importer are not reused.
IMPORTER_VERSION = 4


# ------------------------------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) for one message; length-delimited values are memoryviews."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fnum, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fnum, wt, v


def _packed_varints(v):
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _sint64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


_DTYPES = {1: np.float32, 10: np.float16, 11: np.float64, 6: np.int32, 7: np.int64}


def _tensor(buf):
    dims, dtype, name, raw = [], 1, "", None
    floats, doubles, int32s, int64s = [], [], [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            dims += _packed_varints(v) if wt == 2 else [v]
        elif f == 2:
            dtype = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            floats.append(np.frombuffer(bytes(v), "<f4") if wt == 2 else np.frombuffer(v, "<f4"))
        elif f == 10:
            doubles.append(np.frombuffer(bytes(v), "<f8") if wt == 2 else np.frombuffer(v, "<f8"))
        elif f == 5:
            int32s += _packed_varints(v) if wt == 2 else [v]
        elif f == 7:
            int64s += _packed_varints(v) if wt == 2 else [v]
        elif f == 13 or f == 14:
            pass  # external data / data_location handled by the caller's error below
    shape = [int(_sint64(d)) for d in dims]
    if dtype not in _DTYPES:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, np.dtype(_DTYPES[dtype]).newbyteorder("<"))
    elif floats:
        arr = np.concatenate(floats)
    elif doubles:
        arr = np.concatenate(doubles)
    elif int64s:
        arr = np.array([_sint64(x) for x in int64s], np.int64)
    elif int32s:
        arr = np.array(int32s, np.int32)
        if dtype == 10:  # float16 stored as uint16 bit patterns in int32_data
            arr = arr.astype(np.uint16).view(np.float16)
    else:
        arr = np.zeros(0, _DTYPES[dtype])
    n = int(np.prod(shape)) if shape else arr.size
    if arr.size != n:
        return name, None  # e.g. external data: not supported
    return name, arr.reshape(shape) if shape else arr


def _attr(buf):
    name, val = "", None
    ints, floats = [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _sint64(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5 and wt == 2:      # t: a TensorProto (the `value` of a Constant node)
            val = _tensor(v)[1]
        elif f == 8:
            ints += [_sint64(x) for x in (_packed_varints(v) if wt == 2 else [v])]
        elif f == 7:
            floats.append(np.frombuffer(bytes(v), "<f4") if wt == 2 else np.frombuffer(v, "<f4"))
    if ints:
        val = ints
    elif floats:
        val = list(np.concatenate(floats))
    return name, val


def _node(buf):
    n = dict(op="", name="", inputs=[], outputs=[], attrs={})
    for f, wt, v in _fields(buf):
        if f == 1:
            n["inputs"].append(bytes(v).decode())
        elif f == 2:
            n["outputs"].append(bytes(v).decode())
        elif f == 3:
            n["name"] = bytes(v).decode()
        elif f == 4:
            n["op"] = bytes(v).decode()
        elif f == 5:
            k, val = _attr(v)
            n["attrs"][k] = val
    return n


def _value_info(buf, elem_types=None):
    name, shape = "", []
    for f, wt, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:  # TypeProto
            for f2, _, v2 in _fields(v):
                if f2 == 1:  # tensor_type
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1 and elem_types is not None:  # elem_type (TensorProto.DataType: 1 float, 10 float16)
                            elem_types[name] = int(v3)
                        if f3 == 2:  # shape
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:  # dim
                                    d = -1
                                    for f5, wt5, v5 in _fields(v4):
                                        if f5 == 1:
                                            d = int(_sint64(v5))
                                    shape.append(d)
    return name, shape


class OnnxModel:
    def __init__(self):
        self.initializers = {}
        self.nodes = []
        self.inputs = []   # (name, shape) excluding initializers
        self.outputs = []
        self.elem_types = {}   # graph input / output name -> TensorProto.DataType


def read_onnx(path):
    """Parse an .onnx file -> OnnxModel.  Raises ValueError on anything that is not an ONNX ModelProto."""
    data = memoryview(open(path, "rb").read())
    m = OnnxModel()
    graph = None
    try:
        for f, wt, v in _fields(data):
            if f == 7 and wt == 2:
                graph = v
    except (IndexError, ValueError) as e:
        raise ValueError("[%s] is not an ONNX protobuf: %s" % (path, e))
    if graph is None:
        raise ValueError("[%s] holds no ONNX graph" % path)
    unsupported = []
    for f, wt, v in _fields(graph):
        if f == 1:
            m.nodes.append(_node(v))
        elif f == 5:
            name, arr = _tensor(v)
            if arr is None:
                unsupported.append(name)
            else:
                m.initializers[name] = arr
        elif f == 11:
            m.inputs.append(_value_info(v, m.elem_types))
        elif f == 12:
            m.outputs.append(_value_info(v, m.elem_types))
    # Constant nodes (exporters without constant folding write Reshape shapes, Split sizes, Resize scales, scalar factors this way) are
    # initializers in all but name: fold them, so that every consumer sees one kind of constant
    kept = []
    for nd in m.nodes:
        if nd["op"] == "Constant" and len(nd["outputs"]) == 1:
            a = nd["attrs"]
            val = a.get("value")
            if val is None and "value_float" in a:
                val = np.asarray(a["value_float"], np.float32)
            if val is None and "value_int" in a:
                val = np.asarray(a["value_int"], np.int64)
            if val is None and "value_ints" in a:
                val = np.asarray(a["value_ints"], np.int64)
            if val is None and "value_floats" in a:
                val = np.asarray(a["value_floats"], np.float32)
            if isinstance(val, np.ndarray):
                m.initializers[nd["outputs"][0]] = val
                continue
        kept.append(nd)
    m.nodes = kept
    if unsupported and not m.initializers:
        raise ValueError("[%s]: initializers use external data or unsupported types: %s" % (path, unsupported[:4]))
    m.inputs = [(n, s) for n, s in m.inputs if n not in m.initializers]
    return m


# ------------------------------------------------------------------------------------- architecture detection
def _convs_in_order(m):
    """[(weight fp32 OIHW, bias fp32 | None)] per Conv / ConvTranspose node (the latter's weight in its own (Cin, Cout, kH, kW) layout) in
    graph (= PyTorch execution) order, BatchNormalization nodes that consume a Conv output folded in."""
    by_out = {}
    convs = []
    for nd in m.nodes:
        if nd["op"] in ("Conv", "ConvTranspose") and len(nd["inputs"]) >= 2 and nd["inputs"][1] in m.initializers:
            w = np.asarray(m.initializers[nd["inputs"][1]], np.float32)
            b = np.asarray(m.initializers[nd["inputs"][2]], np.float32) if len(nd["inputs"]) > 2 and nd["inputs"][2] in m.initializers else None
            rec = [w, b, nd]
            convs.append(rec)
            by_out[nd["outputs"][0]] = rec
        elif nd["op"] == "BatchNormalization" and nd["inputs"][0] in by_out:
            rec = by_out[nd["inputs"][0]]
            g, beta, mean, var = (np.asarray(m.initializers[k], np.float32) for k in nd["inputs"][1:5])
            eps = float(nd["attrs"].get("epsilon", 1e-5))
            rec[0], rec[1] = fold_bn(rec[0], rec[1], g, beta, mean, var, eps)
    return [(w, b) for w, b, _ in convs]


def fold_bn(w, b, gamma, beta, mean, var, eps):
    s = gamma / np.sqrt(var + eps)
    w2 = (w * s.reshape(-1, 1, 1, 1)).astype(np.float32)
    b0 = b if b is not None else np.zeros_like(mean)
    return w2, ((b0 - mean) * s + beta).astype(np.float32)


def detect_arch(m):
    """-> (builder name, kwargs).  Raises ValueError with what was found when nothing matches."""
    outs = [s for _, s in m.outputs]
    ins = [s for _, s in m.inputs]
    convs = _convs_in_order(m)
    found = "inputs %s, outputs %s, %d Conv nodes, first conv %s" % (ins, outs, len(convs), convs[0][0].shape if convs else None)
    if not convs or not ins or len(ins[0]) != 4:
        raise ValueError("not a supported architecture: " + found)
    c0 = convs[0][0].shape
    H, W = ins[0][2], ins[0][3]
    if not (isinstance(H, int) and isinstance(W, int) and H > 0 and W > 0):
        raise ValueError("dynamic input size (export with fixed H and W; the engine plans its kernels on static shapes): " + found)
    if len(outs) == 4:                                      # UFLDv2: loc_row, loc_col, exist_row, exist_col
        if c0[2] != 7 or c0[0] != 64:
            raise ValueError("4 outputs but no ResNet stem: " + found)
        depth = {20: "18", 36: "34"}.get(len(convs) - 1)    # + the 1x1 `pool` conv
        if depth is None:
            raise ValueError("ResNet depth not 18/34 (%d convs): %s" % (len(convs), found))
        (_, gr, cr, nl), (_, gc, cc, _) = outs[0], outs[1]
        fc_norm = "cls.0.weight" in m.initializers or any(nd["op"] == "LayerNormalization" for nd in m.nodes)   # Tusimple exports have cls.0 = Identity
        return "ufldv2_res" + depth, dict(in_h=H, in_w=W, num_grid_row=gr, num_cls_row=cr, num_grid_col=gc, num_cls_col=cc, num_lanes=nl,
                                          fc_norm=fc_norm)
    if len(outs) == 1 and len(outs[0]) == 4 and c0[2] == 7 and c0[0] == 64:     # UFLD v1: one (1, G+1, K, L) tensor
        depth = {20: "18", 36: "34"}.get(len(convs) - 1)
        if depth is None:
            raise ValueError("ResNet depth not 18/34 (%d convs): %s" % (len(convs), found))
        _, g1, k, nl = outs[0]
        return "ufld_v1_res" + depth, dict(in_h=H, in_w=W, griding_num=g1 - 1, cls_num_per_lane=k, num_lanes=nl)
    if len(outs) == 1 and len(outs[0]) == 3:
        o = outs[0]
        if c0[2] == 3 and o[2] > o[1]:                      # (1, 4+nc, A): YOLOv8/9/10-style head
            scale = {16: "n", 32: "s", 48: "m", 64: "l", 80: "x"}.get(c0[0])
            if scale is None:
                raise ValueError("YOLOv8 width not supported: " + found)
            if H % 32 or W % 32:
                raise ValueError("YOLO input size must be multiples of 32: " + found)
            # YOLOv10 (the reference's shipped default, demo.py:24-30) exported with the v8-layout head the reference decodes
            # (yoloDetector.py:114,121): same stem and output shape as YOLOv8; told apart by its PSA / one-to-one-head parameters
            # or, when the exporter dropped the names, by its depth-wise convolutions (SCDown, CIB: v8 has none)
            is_v10 = any(".attn.qkv." in k or ".one2one_cv" in k for k in m.initializers) or \
                any(nd["op"] == "Conv" and nd["attrs"].get("group", 1) > 1 for nd in m.nodes)
            if is_v10:
                if scale not in ("n", "s"):
                    raise ValueError("YOLOv10 scale %r is not built (yolov10n and yolov10s are): %s" % (scale, found))
                return "yolov10" + scale, dict(nc=o[1] - 4, imgsz=(H, W))
            # YOLOv9 (GELAN): the same stem width and head as YOLOv8n; its AConv / ADown down-sampling average-pools first (v8 / v10
            # graphs have no AveragePool node), its blocks are RepNCSPELAN4 (parameter names model.N.cv2.0.cv1.conv ...)
            is_v9 = any(nd["op"] == "AveragePool" for nd in m.nodes) or any(".cv2.0.m.0.cv1." in k for k in m.initializers)
            if is_v9:
                v9 = {"n": "t", "s": "s", "l": "c"}.get(scale)      # stem widths 16 / 32 / 64
                if v9 is None:
                    raise ValueError("YOLOv9 with a %d-channel stem is not built (yolov9t, yolov9s and yolov9c are): %s" % (c0[0], found))
                return "yolov9" + v9, dict(nc=o[1] - 4, imgsz=(H, W))
            return "yolov8" + scale, dict(nc=o[1] - 4, imgsz=(H, W))
        if c0[2] == 3 and o[1] > o[2] and any(nd["op"] == "ConvTranspose" for nd in m.nodes):
            # (1, A, 5+nc) with A = one row per cell, transposed-conv up-sampling in the neck: YOLOv6 v3.0 (RepBiFPANNeck); deploy export
            # (RepVGG blocks fused).  Told apart by width: 16-channel stem = n, 32 = s; the conv count pins the depth (0.33).
            if H % 32 or W % 32:
                raise ValueError("YOLO input size must be multiples of 32: " + found)
            scale = {16: "n", 32: "s"}.get(c0[0])
            if scale is None or len(convs) != 71 or o[1] != (H // 8) * (W // 8) + (H // 16) * (W // 16) + (H // 32) * (W // 32):
                raise ValueError("YOLOv6 variant not built (v3.0 yolov6n / yolov6s deploy exports, 69 Conv + 2 ConvTranspose nodes, are): " + found)
            return "yolov6" + scale, dict(nc=o[2] - 5, imgsz=(H, W))
        if c0[2] == 3 and o[1] > o[2]:                      # (1, A, 5+nc) behind a 3x3 stem: YOLOv7 (v5-layout head, yoloDetector.py:110-124)
            if H % 32 or W % 32:
                raise ValueError("YOLO input size must be multiples of 32: " + found)
            n_plain = sum(1 for w_, _ in convs if w_.shape[1] * w_.shape[2] > 0)
            if c0[0] != 32 or n_plain != 58:                # yolov7-tiny: 32-channel stem, 55 Conv modules + the 3 IDetect 1x1s
                raise ValueError("YOLOv7 variant not built (yolov7-tiny, 58 convs behind a 32-channel stem, is): " + found)
            return "yolov7-tiny", dict(nc=o[2] - 5, imgsz=(H, W))
        if c0[2] == 6 and o[1] > o[2]:                      # (1, A, 5+nc): YOLOv5 v6.x
            scale = {16: "n", 32: "s", 48: "m", 64: "l", 80: "x"}.get(c0[0])
            if scale is None:
                raise ValueError("YOLOv5 width not supported: " + found)
            if H % 32 or W % 32:
                raise ValueError("YOLO input size must be multiples of 32: " + found)
            return "yolov5" + scale, dict(nc=o[2] - 5, imgsz=(H, W))
    raise ValueError("not a supported architecture: " + found)


# ------------------------------------------------------------------------------------- weight source for models.build
class OnnxWeights:
    """`wsrc(name, shape, kind, fill=None)` for models.build, backed by an OnnxModel."""

    def __init__(self, m, arch):
        self.m = m
        self.arch = arch
        self.init = m.initializers
        self.convs = _convs_in_order(m)
        self.store = {}
        self.by_name = any(k.endswith(".conv.weight") or k.endswith("conv1.weight") or k.startswith("model.") and k.endswith(".weight")
                           for k in self.init)
        self._order = None

    # conv execution order of a torchvision ResNet as the parsingNet forward visits it (backbone.py:49-58, model_culane.py:48)
    def _resnet_order(self):
        if self._order is None:
            depth = M.RESNET_DEPTHS[self.arch[-2:]]
            names = ["model.conv1"]
            cin = 64
            for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], depth)):
                for bi in range(nblk):
                    s = 2 if (li > 0 and bi == 0) else 1
                    base = "model.layer%d.%d" % (li + 1, bi)
                    names += [base + ".conv1", base + ".conv2"]
                    if s != 1 or cin != planes:
                        names.append(base + ".downsample.0")   # BasicBlock.forward evaluates the shortcut after conv2/bn2
                    cin = planes
            names.append("pool")
            self._order = {n: i for i, n in enumerate(names)}
        return self._order

    def _named_conv(self, base):
        """weight+bias of module `base` (e.g. 'model.0.conv'), BN ('model.0.bn.*' / sibling index) folded when present."""
        w = self.init.get(base + ".weight")
        if w is None:
            return None
        if base.endswith(".conv.conv") and base[:-len(".conv.conv")] + ".conv1.conv.weight" in self.init:
            # an UN-fused RepVGGDW (YOLOv10 CIB with lk=True: 7x7 + 3x3 depth-wise branches, each Conv+BN): re-parameterise to the one
            # 7x7 the deploy form runs (RepVGGDW.fuse: the 3x3 kernel zero-padded to 7x7 and added, biases added)
            stem = base[:-len(".conv.conv")]
            w7, b7 = self._named_conv_plain(stem + ".conv.conv")
            w3, b3 = self._named_conv_plain(stem + ".conv1.conv")
            b7 = np.zeros(w7.shape[0], np.float32) if b7 is None else b7
            b3 = np.zeros(w3.shape[0], np.float32) if b3 is None else b3
            return (w7 + np.pad(w3, ((0, 0), (0, 0), (2, 2), (2, 2)))).astype(np.float32), (b7 + b3).astype(np.float32)
        return self._named_conv_plain(base)

    def _named_conv_plain(self, base):
        w = np.asarray(self.init[base + ".weight"], np.float32)
        b = self.init.get(base + ".bias")
        b = None if b is None else np.asarray(b, np.float32)
        stem = base[:-len(".conv")] if base.endswith(".conv") else None
        bn = stem + ".bn" if stem is not None else None
        if bn is None and ".downsample.0" in base:
            bn = base.replace(".downsample.0", ".downsample.1")
        if bn is None and (base.endswith(".conv1") or base.endswith(".conv2")):
            bn = base[:-5] + "bn" + base[-1]
        if bn is not None and bn + ".running_mean" in self.init:
            g, beta, mean, var = (np.asarray(self.init[bn + k], np.float32) for k in (".weight", ".bias", ".running_mean", ".running_var"))
            eps = 1e-3 if self.arch.startswith("yolo") else 1e-5   # ultralytics Conv BN eps 1e-3, torchvision 1e-5
            w, b = fold_bn(w, b, g, beta, mean, var, eps)
        return w, b

    def _linear_by_order(self, name, shape):
        """Linear weights whose names the exporter dropped: the k-th Gemm/MatMul with a constant operand, transposed when
        stored (in, out)."""
        lin = []
        for nd in self.m.nodes:
            if nd["op"] in ("Gemm", "MatMul") and len(nd["inputs"]) >= 2 and nd["inputs"][1] in self.init:
                w = np.asarray(self.init[nd["inputs"][1]], np.float32)
                if nd["op"] == "MatMul" or not nd["attrs"].get("transB", 0):
                    w = w.T
                lin.append(w)
        order = {"cls.1.weight": 0, "cls.3.weight": 1, "cls.0.weight": 0, "cls.2.weight": 1}.get(name)   # v2 | v1 module indices
        if order is None or order >= len(lin) or lin[order].shape != tuple(shape):
            return None
        return np.ascontiguousarray(lin[order])

    def _v6_by_order(self, name):
        """YOLOv6: upstream module paths differ between releases and exporters drop them; the k-th parameterised layer the builder asks for
        (models.yolov6 requests them in upstream forward order) is the k-th Conv / ConvTranspose node of the export."""
        base = name.rsplit(".", 1)[0]
        if not hasattr(self, "_v6_idx"):
            self._v6_idx = {}
        idx = self._v6_idx.setdefault(base, len(self._v6_idx))      # advanced on EVERY parametrised layer, named in the file or not
        if idx >= len(self.convs):
            return None
        w, b = self.convs[idx]
        return w if name.endswith(".weight") else (b if b is not None else np.zeros(w.shape[1] if w.ndim == 4 and ".upsample_transpose" in base else w.shape[0], np.float32))

    def _v6_lookup(self, name, shape):
        """YOLOv6: the positional tensor of this request, cross-checked against the named one when the export kept names.  A file that
        names only SOME of its layers would let the position fall behind the true layer: refused."""
        pos = self._v6_by_order(name)
        named = self.init.get(name)
        if named is None:
            if not hasattr(self, "_v6_seen_named"):
                self._v6_seen_named = False
            if self._v6_seen_named and name.endswith(".weight"):
                raise ValueError("YOLOv6 export names only part of its layers (%r is missing): cannot map weights by position safely" % name)
            return pos
        if name.endswith(".weight"):
            self._v6_seen_named = True
        if pos is not None and name.endswith(".weight") and pos.size != np.asarray(named).size:
            raise ValueError("YOLOv6 export: tensor %r (%d values) does not sit at its layer's position in the graph (%d values there)" % (name, np.asarray(named).size, pos.size))
        return None     # the named path below takes it

    def __call__(self, name, shape, kind, fill=None):
        if name in self.store:
            return self.store[name]
        arr = None
        if self.arch.startswith("yolov6"):
            arr = self._v6_lookup(name, shape)
        if arr is None and kind in ("conv", "bias") and (name.endswith(".weight") or name.endswith(".bias")):
            base = name.rsplit(".", 1)[0]
            pair = self._named_conv(base) if kind in ("conv", "bias") and base + ".weight" in self.init else None
            if pair is None and self.arch.startswith("ufld"):
                idx = self._resnet_order().get(base)
                if idx is not None and idx < len(self.convs):
                    pair = self.convs[idx]
            if pair is not None:
                w, b = pair
                arr = w if name.endswith(".weight") else (b if b is not None else np.zeros(w.shape[0], np.float32))
        if arr is None and name in self.init:
            arr = np.asarray(self.init[name], np.float32)
        if arr is None and kind == "linear":
            arr = self._linear_by_order(name, shape)
        if arr is None:
            raise KeyError("ONNX file has no tensor for %r (shape %s); known: %s ..." % (name, tuple(shape), sorted(self.init)[:6]))
        arr = np.ascontiguousarray(arr, np.float32).reshape(shape) if arr.size == int(np.prod(shape)) else arr
        if tuple(arr.shape) != tuple(shape):
            raise ValueError("%s: ONNX tensor has shape %s, the %s graph needs %s" % (name, arr.shape, self.arch, tuple(shape)))
        self.store[name] = arr
        return arr


def convert(onnx_path, hipm_path=None, io_half=None):
    """model.onnx -> model.hipm (returns the path and the Graph).

    io_half: None = follow the file (an fp16 export declares float16 graph inputs); True = mark the container as a float16 model
    whatever the file's type -- the counterpart of the reference's onnxQuantization.py:11-41, which rewrites an fp32 ONNX file as
    `<name>_fp16.onnx` with onnxconverter_common so that OnnxEngine reports engine_dtype float16 (coreEngine.py:168).  Here the
    weights stay fp32 in the container (the engine converts them to its compute type at load: `precision="fp16"` rounds them to
    half exactly as the converted file would hold them) and only the I/O contract changes."""
    m = read_onnx(onnx_path)
    try:
        arch, kw = detect_arch(m)
        g = M.build(arch, wsrc=OnnxWeights(m, arch), **kw)
    except (ValueError, KeyError, AssertionError) as e_arch:
        # not one of the hand-built architectures (or a variant of one whose tensors do not fit its builder): lower the node list itself
        # onto the engine's op list (onnx_lower.py) -- another width / depth of a supported family, an ad-hoc CSP graph
        try:
            from . import onnx_lower as OL
        except ImportError:
            import onnx_lower as OL
        try:
            g = OL.lower(m, name=os.path.splitext(os.path.basename(onnx_path))[0][:60])
        except OL.LowerError as e_low:
            raise ValueError("[%s] is neither a hand-built architecture (%s) nor a graph the generic lowering takes (%s)" % (onnx_path, e_arch, e_low))
    # an fp16 export (onnxQuantization.py:11-41 / ultralytics half=True) declares float16 graph inputs: the reference then feeds and
    # receives float16 arrays (coreEngine.py:168); recorded in the container so HipEngine can report the same engine_dtype
    g.io_half = (bool(m.inputs) and m.elem_types.get(m.inputs[0][0]) == 10) if io_half is None else bool(io_half)
    hipm_path = hipm_path or os.path.splitext(onnx_path)[0] + ".hipm"
    g.save(hipm_path)
    return hipm_path, g


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if a != "--half"]
    half = True if "--half" in sys.argv[1:] else None
    if len(argv) < 1:
        raise SystemExit("usage: python onnx_import.py [--half] model.onnx [model.hipm]\n"
                         "  --half  write `<name>_fp16.hipm`: a float16-I/O model (what onnxQuantization.py produces as <name>_fp16.onnx)")
    dst = argv[1] if len(argv) > 1 else (os.path.splitext(argv[0])[0] + ("_fp16" if half else "") + ".hipm")
    p, g = convert(argv[0], dst, io_half=half)
    print("%s: %s, %d convs, %.2f GFLOP/frame, %.2f M parameters%s" % (p, g.name, g.n_convs, g.flops / 1e9, g.n_params / 1e6,
                                                                       ", float16 I/O" if g.io_half else ""))
