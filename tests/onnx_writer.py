"""Minimal ONNX (protobuf) WRITER for tests: emits files that follow the naming / ordering conventions of the two
exporters the reference relies on, so the dependency-free reader in vehicle-cv-adas_amd/onnx_import.py can be exercised
without the `onnx` package.  Test infrastructure only."""
import struct
import numpy as np


def _varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, value):
    return _varint((field << 3) | 0) + _varint(value)


def tensor(name, arr, raw=True, dtype=None):
    arr = np.ascontiguousarray(arr)
    code = {np.dtype(np.float32): 1, np.dtype(np.float16): 10, np.dtype(np.float64): 11, np.dtype(np.int64): 7}[arr.dtype]
    out = b"".join(_vi(1, d) for d in arr.shape) + _vi(2, code) + _ld(8, name.encode())
    if raw:
        out += _ld(9, arr.tobytes())
    else:
        assert arr.dtype == np.float32
        out += _ld(4, arr.astype("<f4").tobytes())       # packed float_data
    return out


def attr_ints(name, ints):
    return _ld(1, name.encode()) + _ld(8, b"".join(_varint(i) for i in ints)) + _vi(20, 7)


def attr_int(name, i):
    return _ld(1, name.encode()) + _vi(3, i) + _vi(20, 2)


def attr_float(name, f):
    return _ld(1, name.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", f) + _vi(20, 1)


def attr_floats(name, vals):
    return _ld(1, name.encode()) + _ld(7, struct.pack("<%df" % len(vals), *vals)) + _vi(20, 6)


def attr_str(name, text):
    return _ld(1, name.encode()) + _ld(4, text.encode()) + _vi(20, 3)


def attr_tensor(name, arr, tname=""):
    return _ld(1, name.encode()) + _ld(5, tensor(tname, arr)) + _vi(20, 4)


def node(op, inputs, outputs, name="", attrs=()):
    out = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    out += _ld(3, name.encode()) + _ld(4, op.encode()) + b"".join(_ld(5, a) for a in attrs)
    return out


def value_info(name, shape, elem_type=1):
    dims = b"".join(_ld(1, _vi(1, d)) for d in shape)
    ttype = _vi(1, elem_type) + _ld(2, dims)
    return _ld(1, name.encode()) + _ld(2, _ld(1, ttype))


def model(nodes, initializers, inputs, outputs, elem_type=1):
    """elem_type: TensorProto.DataType of the graph inputs/outputs (1 float32, 10 float16 = an fp16 export)."""
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in initializers)
    g += b"".join(_ld(11, value_info(n, s, elem_type)) for n, s in inputs) + b"".join(_ld(12, value_info(n, s, elem_type)) for n, s in outputs)
    return _vi(1, 8) + _ld(2, b"adas-hip-tests") + _ld(7, g)
