"""GPU: an ONNX file of a graph models.py does not hand-build goes straight into HipEngine (coreEngine.py:159-186: the reference hands any
ONNX file to its engine): the generic lowering (onnx_lower.py) turns it into the engine's op list, the engine's load-time passes fuse it
like a hand-built graph, and the outputs match the torch interpreter of the ORIGINAL op list (tests/graph_interp.py) within the
precisions' usual bounds."""
import importlib

import numpy as np
import pytest

import graph_interp
import onnx_emit
from conftest import load_pkg

pytestmark = pytest.mark.gpu
load_pkg()
M = importlib.import_module("adas_amd.models")
CE = importlib.import_module("adas_amd.coreEngine")


def _custom_v8(tag, depth, width, **kw):
    M.V8_SCALES[tag] = (depth, width, 1024)
    try:
        return M.yolov8(tag, **kw)
    finally:
        del M.V8_SCALES[tag]


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("fp16x3", 1e-5), ("fp16", 5e-3), ("bf16", 4e-2)])
def test_custom_scale_yolov8_onnx_runs_through_hipengine(tmp_path, prec, tol):
    g = _custom_v8("q", 0.67, 0.375, imgsz=(128, 160), nc=80)          # width 0.375: 24-channel stem -- no builder, no detect_arch entry
    path = str(tmp_path / "yolov8q.onnx")
    onnx_emit.emit(g, path)
    x = np.random.default_rng(1).uniform(0, 1, (2, 3, 128, 160)).astype(np.float32)
    want = graph_interp.run(g, x)[0]
    e = CE.OnnxEngine(path, precision=prec, max_batch=2)               # the reference's call: OnnxEngine("model.onnx")
    assert e.get_engine_input_shape() == [1, 3, 128, 160] and e.get_engine_output_shape()[0] == [[1, 84, want.shape[2]]]
    got = np.array(e.engine_inference(x)[0], copy=True)
    kernels = [e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])]
    e.close()
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    print("lowered yolov8(custom) %s: rel %.2e  max|prob diff| %.2e" % (prec, rel, float(np.abs(got[:, 4:] - want[:, 4:]).max())))
    assert rel <= tol
    if prec == "fp16":      # the engine's fusion passes see a lowered graph like a hand-built one (the upsample fold needs a K-step count
        assert any("detect_v8_fused_kernel" in k for k in kernels), kernels   # conv_pw instantiates: 576 / 288 input channels here are not)
    if prec == "fp16x3":    # ... and in the exact mode: the split-precision Detect fusion (round 6)
        assert any("detect_v8_fused_x3_kernel" in k for k in kernels), kernels


@pytest.mark.parametrize("nc", [16, 24, 96, 11])
def test_fused_x3_detect_with_other_class_counts(tmp_path, monkeypatch, nc):
    """detect_v8_fused_x3_kernel with class counts that are not 80: 16 (one class tile), 24 (two, the second half wide), 96 (six) -- the head
    against the torch interpreter and against the same engine with the fusion off.  11 classes: the class branch's closing conv is not a
    multiple of 8 channels wide and stays on the separate launches (the head must still be right)."""
    g = M.yolov8("n", imgsz=(96, 160), nc=nc)
    path = str(tmp_path / ("yolov8n_nc%d.onnx" % nc))
    onnx_emit.emit(g, path)
    x = np.random.default_rng(2).uniform(0, 1, (3, 3, 96, 160)).astype(np.float32)
    want = graph_interp.run(g, x)[0]
    e = CE.OnnxEngine(path, precision="fp16x3", max_batch=3)
    got = np.array(e.engine_inference(x)[0], copy=True)
    fused = any("detect_v8_fused_x3_kernel" in e.layer_kernel(i, 3) for i in range(e.stats()["num_layers"]))
    e.close()
    monkeypatch.setenv("ADAS_NO_DETECT_FUSE", "1")
    e = CE.OnnxEngine(path, precision="fp16x3", max_batch=3)
    sep = np.array(e.engine_inference(x)[0], copy=True)
    e.close()
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    dp, db = float(np.abs(got[:, 4:] - sep[:, 4:]).max()), float(np.abs(got[:, :4] - sep[:, :4]).max())
    print("fused x3 Detect nc=%d: rel %.2e vs interpreter; vs separate launches max|prob diff| %.2e max|box diff| %.2e px" % (nc, rel, dp, db))
    assert fused == (nc % 8 == 0) and got.shape == (3, 4 + nc, want.shape[2]), (fused, got.shape)
    assert rel <= 1e-5 and dp <= 2e-6 and db <= 2e-3


def test_v5_layout_onnx_runs_through_hipengine(tmp_path):
    g = M.build("yolov7-tiny", imgsz=(96, 128), nc=11)
    path = str(tmp_path / "v7_generic.onnx")
    onnx_emit.emit(g, path)
    OL = importlib.import_module("adas_amd.onnx_lower")
    OI = importlib.import_module("adas_amd.onnx_import")
    g2 = OL.lower(OI.read_onnx(path), "v7g")                           # the generic path (convert() would pick the hand-built builder)
    hipm = str(tmp_path / "v7_generic.hipm")
    g2.save(hipm)
    x = np.random.default_rng(2).uniform(0, 1, (2, 3, 96, 128)).astype(np.float32)
    want = graph_interp.run(g, x)[0]
    e = CE.HipEngine(hipm, precision="fp32", max_batch=2)
    got = np.array(e.engine_inference(x)[0], copy=True)
    e.close()
    assert got.shape == want.shape
    assert float(np.abs(got - want).max()) <= 1e-3 * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("fp16x3", 1e-5), ("fp16", 5e-3)])
def test_bifpn_style_sums_onnx_runs_through_hipengine(tmp_path, prec, tol):
    """A graph with stand-alone sums of feature maps (weighted, up-sampled terms, swish / ReLU behind them: BiFPN / CBFuse style) as an
    exporter writes it -> OnnxEngine: the sums run as wsum_kernel launches with the up-sampling folded into their loads."""
    from test_onnx_lower import fuse_graph
    g = fuse_graph(128)
    path = str(tmp_path / "fuse.onnx")
    onnx_emit.emit(g, path)
    x = np.random.default_rng(3).uniform(0, 1, (2, 3, 128, 128)).astype(np.float32)
    want = graph_interp.run(g, x)[0]
    e = CE.OnnxEngine(path, precision=prec, max_batch=2)
    got = np.array(e.engine_inference(x)[0], copy=True)
    kernels = [e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])]
    e.close()
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    print("lowered sums graph %s: rel %.2e" % (prec, rel))
    assert rel <= tol and kernels.count("wsum_kernel") == 5 and "upsample2_kernel" not in kernels


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("fp16x3", 1e-5), ("fp16", 5e-3)])
def test_custom_scale_yolov10_onnx_with_attention_runs_through_hipengine(tmp_path, prec, tol):
    """OnnxEngine on a YOLOv10 of a scale outside models.py (3-head PSA): the lowered attention runs on attention_mfma_kernel / attention_kernel."""
    from test_onnx_lower import custom_v10
    g = custom_v10(imgsz=(128, 160), nc=80)
    path = str(tmp_path / "yolov10q.onnx")
    onnx_emit.emit(g, path)
    x = np.random.default_rng(4).uniform(0, 1, (2, 3, 128, 160)).astype(np.float32)
    want = graph_interp.run(g, x)[0]
    e = CE.OnnxEngine(path, precision=prec, max_batch=2)
    got = np.array(e.engine_inference(x)[0], copy=True)
    kernels = [e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])]
    e.close()
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    print("lowered yolov10(custom) %s: rel %.2e" % (prec, rel))
    assert rel <= tol and any("attention" in k for k in kernels)


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("fp16x3", 1e-5), ("fp16", 5e-3), ("bf16", 4e-2)])
def test_shufflenet_onnx_runs_through_hipengine(tmp_path, prec, tol):
    """ShuffleNetV2 units (YOLOv5-lite style backbone) through OnnxEngine: the channel shuffle runs as shuffle_kernel in every storage type."""
    from test_onnx_lower import shufflenet_graph
    g = shufflenet_graph(128)
    path = str(tmp_path / "shuffle.onnx")
    onnx_emit.emit(g, path)
    x = np.random.default_rng(5).uniform(0, 1, (2, 3, 128, 128)).astype(np.float32)
    want = graph_interp.run(g, x)[0]
    e = CE.OnnxEngine(path, precision=prec, max_batch=2)
    got = np.array(e.engine_inference(x)[0], copy=True)
    kernels = [e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])]
    e.close()
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    print("lowered shufflenet %s: rel %.2e" % (prec, rel))
    assert rel <= tol and kernels.count("shuffle_kernel") == 8


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("fp16x3", 1e-5), ("fp16", 5e-3), ("bf16", 4e-2)])
def test_hard_swish_network_onnx_runs_through_hipengine(tmp_path, prec, tol):
    """A PP-LCNet-style graph (hard-swish everywhere, ReLU / hard-sigmoid squeeze-and-excitation, a bare hard-sigmoid: the YOLOv5-lite-c
    family) as an exporter writes it -> OnnxEngine: every hard-swish runs as a one-input wsum_kernel launch behind an activation-free
    convolution, the gates through se_gate_kernel's ReLU / hard-sigmoid forms -- against the torch interpreter of the original graph."""
    from test_onnx_lower import lcnet_graph
    g = lcnet_graph(128)
    path = str(tmp_path / "lcnet.onnx")
    onnx_emit.emit(g, path, hswish_as_mul=(prec == "fp16x3"))
    x = np.random.default_rng(5).uniform(0, 1, (3, 3, 128, 128)).astype(np.float32)
    want = graph_interp.run(g, x)[0]
    e = CE.OnnxEngine(path, precision=prec, max_batch=3)
    got = np.array(e.engine_inference(x)[0], copy=True)
    kernels = [e.layer_kernel(i, 3) for i in range(e.stats()["num_layers"])]
    e.close()
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    print("lowered hard-swish graph %s: rel %.2e  max|prob diff| %.2e" % (prec, rel, float(np.abs(got[:, 4:] - want[:, 4:]).max())))
    assert rel <= tol
    assert sum("wsum_kernel" in k for k in kernels) == 14 and sum("se_gate" in k for k in kernels) == 2, kernels      # 11 hard-swish, 1 hard-sigmoid, 2 ReLU6


def test_hard_swish_is_not_a_convolution_epilogue(tmp_path):
    """The conv kernels apply SiLU / ReLU / LeakyReLU only: a container whose convolution asks for hard-swish is refused at load time,
    naming the layer -- never run with the activation silently dropped."""
    ws = M.SynthWeights(1, gain=1.0)
    g = M.Graph("bad", 3, 64, 64, ws)
    x, cin = g.input()
    y = g.conv(x, 16, 3, 2, "stem", act=M.ACT_HSWISH, true_cin=cin)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = str(tmp_path / "bad.hipm")
    g.save(path)
    with pytest.raises(Exception, match="not a convolution epilogue"):
        CE.HipEngine(path, "fp16", 1)

