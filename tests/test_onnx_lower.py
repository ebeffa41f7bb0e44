"""CPU: the ONNX graph lowering (vehicle-cv-adas_amd/onnx_lower.py): detector graphs that models.py does NOT hand-build are written as ONNX
files in the exporters' node vocabulary (tests/onnx_emit.py), lowered back onto the engine's op list and run through the torch interpreter
of that op list (tests/graph_interp.py) next to the original graph: same outputs, no extra copies where producers can write into the
concat buffers.  The reference loads ANY ONNX file through ONNXRuntime (coreEngine.py:159-186); this is the counterpart for graphs outside
the hand-built set."""
import importlib
import os

import numpy as np
import pytest

import graph_interp
import onnx_emit
import onnx_writer as OW
from conftest import load_pkg

load_pkg()
M = importlib.import_module("adas_amd.models")
OI = importlib.import_module("adas_amd.onnx_import")
OL = importlib.import_module("adas_amd.onnx_lower")


def _roundtrip(g, path, use_split=True, n=2, via_convert=False):
    onnx_emit.emit(g, str(path), use_split)
    if via_convert:
        hipm, g2 = OI.convert(str(path), str(path) + ".hipm")
        assert os.path.getsize(hipm) > 0
    else:
        g2 = OL.lower(OI.read_onnx(str(path)), "t")
    x = np.random.default_rng(0).uniform(0, 1, (n, 3, g.in_h, g.in_w)).astype(np.float32)
    a, b = graph_interp.run(g, x), graph_interp.run(g2, x)
    assert len(a) == len(b) == 1 and a[0].shape == b[0].shape
    return g2, float(np.abs(a[0] - b[0]).max())


def _custom_v8(tag, depth, width, **kw):
    M.V8_SCALES[tag] = (depth, width, 1024)
    try:
        return M.yolov8(tag, **kw)
    finally:
        del M.V8_SCALES[tag]


@pytest.mark.parametrize("use_split", [True, False], ids=["Split", "Slice"])
def test_yolov8_custom_scale_not_in_models(tmp_path, use_split):
    """A YOLOv8 at width 0.375 / depth 0.67 (24-channel stem, two Bottlenecks per C2f): no builder has it, detect_arch rejects it, the
    lowering maps it op for op -- same op count, same buffer count (every Concat input is written in place), identical outputs."""
    g = _custom_v8("q", 0.67, 0.375, imgsz=(64, 96), nc=12)
    with pytest.raises(ValueError):
        onnx_emit.emit(g, str(tmp_path / "q0.onnx"), use_split)
        OI.detect_arch(OI.read_onnx(str(tmp_path / "q0.onnx")))
    g2, err = _roundtrip(g, tmp_path / "q.onnx", use_split, via_convert=True)
    assert err == 0.0
    assert len(g2.ops) == len(g.ops) and len(g2.bufs) == len(g.bufs)
    assert g2.meta["kind"] == "yolov8" and g2.meta["nc"] == 12
    assert g2.outs[0][2] == g.outs[0][2]


def test_v5_layout_head_and_leaky_relu(tmp_path):
    """YOLOv7-tiny's op vocabulary (LeakyReLU, SPP max-pools of one tensor, IDetect in deploy form) through the GENERIC path: the anchors
    come out of the graph's anchor_grid constants."""
    g = M.build("yolov7-tiny", imgsz=64, nc=7)
    g2, err = _roundtrip(g, tmp_path / "v7.onnx")
    assert err == 0.0 and g2.meta["kind"] == "yolov5" and len(g2.ops) == len(g.ops)
    d0 = [o for o in g.ops if o["type"] == M.OP_DETECT_V5][0]
    d1 = [o for o in g2.ops if o["type"] == M.OP_DETECT_V5][0]
    blob0, blob1 = np.frombuffer(bytes(g.blob), np.float32), np.frombuffer(bytes(g2.blob), np.float32)
    assert np.array_equal(blob0[d0["w"][0] // 4: d0["w"][0] // 4 + 18], blob1[d1["w"][0] // 4: d1["w"][0] // 4 + 18])


def test_gelan_style_graph_with_average_pools(tmp_path):
    """YOLOv9t's vocabulary (AConv average pools, RepNCSPELAN4 with nested splits and concats): tensors that sit in two concats are
    copied into the second one (a 1x1 max-pool), everything else is written in place."""
    g = M.build("yolov9t", imgsz=64, nc=5)
    g2, err = _roundtrip(g, tmp_path / "v9.onnx")
    assert err == 0.0
    copies = [o for o in g2.ops if o["type"] == M.OP_MAXPOOL and o["kh"] == 1]
    assert len(g2.ops) - len(g.ops) == len(copies) <= 16


def test_adhoc_graph_residuals_both_orders_and_depthwise(tmp_path):
    """A graph no family has: ResNet-style pre-activation residual (Conv -> Add -> ReLU), Bottleneck-style post-activation residual,
    a depth-wise 3x3, nested concats, and a v8 Detect head on three ad-hoc pyramid levels."""
    ws = M.SynthWeights(3, gain=1.0)
    g = M.Graph("adhoc", 3, 128, 128, ws)
    x, cin = g.input()
    x = g.conv(x, 16, 3, 2, "stem", act=M.ACT_RELU, true_cin=cin)
    x = g.maxpool(x, 2, 2, 0, name="stem.pool")                                                # pyramid levels at stride 8 / 16 / 32
    t = g.conv(x, 16, 3, 1, "b0.conv1", act=M.ACT_RELU)
    x = g.conv(t, 16, 3, 1, "b0.conv2", act=M.ACT_RELU, res=x, res_mode=M.RES_BEFORE_ACT)
    cat = g.buf(16, 16, 48)
    p3 = g.conv(x, 32, 3, 2, "down1", out=cat.slice(0, 32))
    g.dwconv(p3.slice(0, 16), 3, 1, "dw", act=M.ACT_SILU, out=cat.slice(32, 16), res=p3.slice(16, 16))
    p3 = g.conv(cat, 32, 1, 1, "mix", act=M.ACT_LEAKY)
    p4 = g.conv(p3, 64, 3, 2, "down2")
    t = g.conv(p4, 64, 3, 1, "b1.cv1")
    p4 = g.conv(t, 64, 3, 1, "b1.cv2", res=p4, res_mode=M.RES_AFTER_ACT)
    p5 = g.maxpool(p4, 2, 2, 0, name="mp")
    ins, strides, nc = [], [], 9
    for i, f in enumerate((p3, p4, p5)):
        b = g.conv(f, 64, 1, 1, "head.cv2.%d" % i, act=M.ACT_NONE, f32_out=True)
        c = g.conv(f, nc, 1, 1, "head.cv3.%d" % i, act=M.ACT_NONE, f32_out=True)
        ins += [b, c]
        strides.append(128 // f.h)
    A = sum(f.h * f.w for f in (p3, p4, p5))
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(M.OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    g2, err = _roundtrip(g, tmp_path / "adhoc.onnx", via_convert=True)
    assert err == 0.0 and len(g2.ops) == len(g.ops)
    kinds = [(o["type"], o["act"], o["res_mode"]) for o in g2.ops]
    assert (M.OP_CONV, M.ACT_RELU, M.RES_BEFORE_ACT) in kinds and (M.OP_CONV, M.ACT_SILU, M.RES_AFTER_ACT) in kinds
    assert any(o["type"] == M.OP_DWCONV and o["res_mode"] == M.RES_AFTER_ACT for o in g2.ops)


def test_standalone_sums_become_weighted_sum_ops(tmp_path):
    """BiFPN / CBFuse style nodes -- a sum of 2-3 feature maps, terms scaled by constants, a nearest x2 Resize in front of a term, swish or
    ReLU behind the sum -- written the way exporters write them (Resize, Mul by a scalar, a chain of Adds, Sigmoid * Mul) come back as ONE
    weighted-sum op each (the up-sampling folded into its loads), next to convolution residuals, which stay inside their convolutions."""
    g = fuse_graph()
    g2, err = _roundtrip(g, tmp_path / "fuse.onnx", via_convert=True)
    assert err == 0.0 and len(g2.ops) == len(g.ops)
    sums = [o for o in g2.ops if o["type"] == M.OP_WSUM]
    assert [len(o["ins"]) for o in sums] == [2, 2, 3, 2, 2] and [o["act"] for o in sums] == [M.ACT_SILU, M.ACT_NONE, M.ACT_SILU, M.ACT_RELU, M.ACT_LEAKY]
    assert [o["ins"][1].h * 2 == o["out"].h for o in sums] == [True, True, False, False, False]              # folded up-sampling where the graph had a Resize
    np.testing.assert_allclose(sums[2]["params"][:3], M.fusion_weights([0.9, 1.1, 0.4]), rtol=0, atol=0)
    assert not any(o["type"] == M.OP_UPSAMPLE2 for o in g2.ops)
    assert any(o["type"] == M.OP_CONV and o["res_mode"] == M.RES_AFTER_ACT for o in g2.ops)


def shufflenet_graph(hw=128):
    """ShuffleNetV2 units (the YOLOv5-lite backbones of the reference's model table): stride-1 units (chunk, right branch 1x1 -> depth-wise 3x3
    -> 1x1, concat, channel shuffle) and stride-2 units (both branches down-sample), ReLU, a max-pooled stem, v8-layout head."""
    ws = M.SynthWeights(9, gain=1.0)
    g = M.Graph("shuffle", 3, hw, hw, ws)
    x, cin = g.input()
    x = g.conv(x, 32, 3, 2, "stem.conv", act=M.ACT_RELU, true_cin=cin)
    x = g.maxpool(x, 3, 2, 1, name="stem.pool")                                  # stride 4

    def unit(x, cout, stride, nm):
        half = cout // 2
        cat = g.buf(x.h // stride, x.w // stride, cout)
        if stride == 1:
            g.maxpool(x.slice(0, half), 1, 1, 0, out=cat.slice(0, half), name=nm + ".keep")      # x1 passes through (a channel copy)
            r = x.slice(half, half)
        else:
            t = g.dwconv(x, 3, 2, nm + ".b1.dw", act=M.ACT_NONE)
            g.conv(t, half, 1, 1, nm + ".b1.pw", act=M.ACT_RELU, out=cat.slice(0, half))
            r = x
        t = g.conv(r, half, 1, 1, nm + ".b2.pw1", act=M.ACT_RELU)
        t = g.dwconv(t, 3, stride, nm + ".b2.dw", act=M.ACT_NONE)
        g.conv(t, half, 1, 1, nm + ".b2.pw2", act=M.ACT_RELU, out=cat.slice(half, half))
        return g.shuffle(cat, 2, nm + ".shuffle")

    feats = []
    for si, (c, n) in enumerate(((64, 2), (128, 2), (256, 1))):
        x = unit(x, c, 2, "s%d.0" % si)
        for r in range(n):
            x = unit(x, c, 1, "s%d.%d" % (si, r + 1))
        feats.append(x)
    ins, strides, nc = [], [], 8
    for i, f in enumerate(feats):
        ins += [g.conv(f, 64, 1, 1, "head.cv2.%d" % i, act=M.ACT_NONE, f32_out=True), g.conv(f, nc, 1, 1, "head.cv3.%d" % i, act=M.ACT_NONE, f32_out=True)]
        strides.append(hw // f.h)
    A = sum(f.h * f.w for f in feats)
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(M.OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    return g


def test_shufflenet_units_with_channel_shuffle(tmp_path):
    """Reshape (B, 2, C / 2, H, W) -> Transpose -> Reshape = torch channel_shuffle comes back as one shuffle op; chunk / concat stay views."""
    g = shufflenet_graph()
    g2, err = _roundtrip(g, tmp_path / "shuffle.onnx", via_convert=True)
    assert err == 0.0
    assert sum(o["type"] == M.OP_SHUFFLE for o in g2.ops) == sum(o["type"] == M.OP_SHUFFLE for o in g.ops) == 8
    assert all(int(o["params"][0]) == 2 for o in g2.ops if o["type"] == M.OP_SHUFFLE) and len(g2.ops) <= len(g.ops)


def custom_v10(tag="q", depth=0.67, width=0.375, **kw):
    M.V10_SCALES[tag] = (depth, width, 1024)
    try:
        return M.yolov10(tag, **kw)
    finally:
        del M.V10_SCALES[tag]


def test_yolov10_custom_scale_with_psa_attention(tmp_path):
    """A YOLOv10 of a scale models.py does not build (width 0.375: a 3-head PSA) as ultralytics exports it: the softmax attention
    (Reshape -> Split -> q^T k * scale -> Softmax -> v attn^T -> Reshape, + the depth-wise `pe` conv of v, Add) comes back as one attention op
    and one depth-wise conv per head; SCDown / CIB depth-wise convs, the RepVGGDW 7x7 and the one-to-one v8-layout head like any other graph."""
    g = custom_v10(imgsz=(128, 160), nc=80)
    g2, err = _roundtrip(g, tmp_path / "v10q.onnx", via_convert=True)
    att = [o for o in g2.ops if o["type"] == M.OP_ATTENTION]
    assert err == 0.0 and len(att) == 1 and [int(v) for v in att[0]["params"][:3]] == [3, 32, 64]
    assert len(g2.ops) <= len(g.ops) + 1            # the builder writes b + ffn(b) into b's slot of the (a, b) buffer; the lowering keeps one copy
    assert sum(o["type"] == M.OP_DWCONV and o["res"] is not None and o["res"].buf == att[0]["out"].buf for o in g2.ops) == 3


def test_mbconv_blocks_with_squeeze_and_excitation(tmp_path):
    """EfficientNet-style inverted-residual blocks (1x1 expand, depth-wise 3x3 / 5x5, swish squeeze-and-excitation, linear 1x1 project with
    an identity skip) as torch exports them -- GlobalAveragePool -> Conv -> Sigmoid * Mul -> Conv -> Sigmoid -> Mul(x, gate) -- come back as
    the gate + scale operator pair; the skip stays inside the project convolution."""
    ws = M.SynthWeights(6, gain=1.0)
    g = M.Graph("mb", 3, 128, 128, ws)
    x, cin = g.input()
    x = g.conv(x, 16, 3, 2, "stem", true_cin=cin)
    x = M._mbconv(g, x, "blocks.0", 1, 3, 1, 16)           # expand ratio 1, skip
    x = M._mbconv(g, x, "blocks.1", 6, 3, 2, 24)
    p3 = M._mbconv(g, x, "blocks.2", 6, 5, 2, 40)           # stride 8
    t = M._mbconv(g, p3, "blocks.3", 6, 5, 1, 40)           # skip
    p4 = M._mbconv(g, t, "blocks.4", 6, 3, 2, 80)
    p5 = M._mbconv(g, p4, "blocks.5", 6, 5, 2, 112)
    ins, strides, nc = [], [], 8
    for i, f in enumerate((t, p4, p5)):
        ins += [g.conv(f, 64, 1, 1, "head.cv2.%d" % i, act=M.ACT_NONE, f32_out=True), g.conv(f, nc, 1, 1, "head.cv3.%d" % i, act=M.ACT_NONE, f32_out=True)]
        strides.append(128 // f.h)
    A = sum(f.h * f.w for f in (t, p4, p5))
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(M.OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    g2, err = _roundtrip(g, tmp_path / "mb.onnx", via_convert=True)
    assert err == 0.0 and len(g2.ops) == len(g.ops)
    assert sum(o["type"] == M.OP_SE_GATE for o in g2.ops) == 6 and sum(o["type"] == M.OP_SCALE for o in g2.ops) == 6
    assert [int(o["params"][0]) for o in g2.ops if o["type"] == M.OP_SE_GATE] == [4, 4, 6, 10, 10, 20]       # squeeze widths = 0.25 x block input
    assert sum(o["type"] == M.OP_CONV and o["res_mode"] == M.RES_AFTER_ACT for o in g2.ops) == 2


def lcnet_graph(hw=128):
    """PP-LCNet-style backbone (the YOLOv5-lite-c family of the reference's model table): a hard-swish stem, depth-separable blocks
    (depth-wise 3x3 / 5x5 -> hard-swish -> [squeeze-and-excitation with ReLU inside and a hard-sigmoid gate] -> 1x1 -> hard-swish), a
    stand-alone hard-sigmoid, v8-layout head.  The convolutions carry no activation: every hard-swish is its own element-wise layer."""
    ws = M.SynthWeights(12, gain=1.0)
    g = M.Graph("lcnet", 3, hw, hw, ws)
    x, cin = g.input()
    x = g.act(g.conv(x, 16, 3, 2, "stem.conv", act=M.ACT_NONE, true_cin=cin), M.ACT_HSWISH, "stem.act")

    def block(x, cout, k, s, se, nm):
        t = g.act(g.dwconv(x, k, s, nm + ".dw", act=M.ACT_NONE), M.ACT_HSWISH, nm + ".dw_act")
        if se:
            t = g.se(t, max(8, t.c // 4), nm + ".se", hidden_act=M.ACT_RELU, gate_act=M.ACT_HSIGMOID)
        return g.act(g.conv(t, cout, 1, 1, nm + ".pw", act=M.ACT_NONE), M.ACT_HSWISH, nm + ".pw_act")

    x = block(x, 32, 3, 1, False, "b1")
    t = g.act(g.conv(x, 96, 1, 1, "ir.expand", act=M.ACT_NONE), M.ACT_RELU6, "ir.expand_act")          # a MobileNetV2 inverted residual: ReLU6
    t = g.act(g.dwconv(t, 3, 1, "ir.dw", act=M.ACT_NONE), M.ACT_RELU6, "ir.dw_act")
    x = g.conv(t, 32, 1, 1, "ir.project", act=M.ACT_NONE, res=x, res_mode=M.RES_AFTER_ACT)
    x = block(x, 64, 3, 2, False, "b2")
    p3 = block(x, 128, 3, 2, False, "b3")            # stride 8
    p4 = block(p3, 256, 5, 2, True, "b4")            # stride 16, SE
    p5 = block(p4, 256, 5, 2, True, "b5")            # stride 32, SE
    p5 = g.conv(p5, 256, 1, 1, "neck.mix", act=M.ACT_NONE, res=g.act(p5, M.ACT_HSIGMOID, "neck.gate"), res_mode=M.RES_AFTER_ACT)   # a bare hard-sigmoid
    ins, strides, nc = [], [], 8
    for i, f in enumerate((p3, p4, p5)):
        ins += [g.conv(f, 64, 1, 1, "head.cv2.%d" % i, act=M.ACT_NONE, f32_out=True), g.conv(f, nc, 1, 1, "head.cv3.%d" % i, act=M.ACT_NONE, f32_out=True)]
        strides.append(hw // f.h)
    A = sum(f.h * f.w for f in (p3, p4, p5))
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(M.OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    return g


@pytest.mark.parametrize("as_mul", [False, True], ids=["HardSwish", "x*HardSigmoid"])
def test_hard_swish_network_lowers_to_elementwise_activation_layers(tmp_path, as_mul):
    """torch.nn.Hardswish / Hardsigmoid / ReLU6 (MobileNetV3 / PP-LCNet / MobileNetV2 networks: YOLOv5-lite-c) have no conv epilogue: a HardSwish node -- or the
    x * HardSigmoid(alpha 1/6) pair older opsets write -- behind a convolution becomes a ONE-input weighted-sum layer, the convolution
    keeps ACT_NONE; the ReLU / hard-sigmoid squeeze-and-excitation becomes the gate op with its two activation codes."""
    g = lcnet_graph()
    path = tmp_path / "lcnet.onnx"
    onnx_emit.emit(g, str(path), hswish_as_mul=as_mul)
    m = OI.read_onnx(str(path))
    ops = {nd["op"] for nd in m.nodes}
    assert ("HardSwish" in ops) == (not as_mul) and "HardSigmoid" in ops
    g2 = OL.lower(m, "t")
    x = np.random.default_rng(0).uniform(0, 1, (2, 3, g.in_h, g.in_w)).astype(np.float32)
    a, b = graph_interp.run(g, x)[0], graph_interp.run(g2, x)[0]
    assert np.array_equal(a, b) and len(g2.ops) == len(g.ops)
    acts = [o for o in g2.ops if o["type"] == M.OP_WSUM]
    assert sum(o["act"] == M.ACT_HSWISH for o in acts) == 11 and sum(o["act"] == M.ACT_HSIGMOID for o in acts) == 1 and all(len(o["ins"]) == 1 for o in acts)
    assert sum(o["act"] == M.ACT_RELU6 for o in acts) == 2          # Clip(x, 0, 6), bounds as inputs or (older exports) as attributes
    gates = [o for o in g2.ops if o["type"] == M.OP_SE_GATE]
    assert [(int(o["params"][1]), int(o["params"][2])) for o in gates] == [(M.ACT_RELU, M.ACT_HSIGMOID)] * 2
    assert all(o["act"] <= M.ACT_LEAKY for o in g2.ops if o["type"] in (M.OP_CONV, M.OP_DWCONV))
    # ONNX's DEFAULT HardSigmoid (alpha 0.2) is another function: refused, not approximated
    nodes = [OW.node("Conv", ["images", "w"], ["c"], "/c", [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_ints("pads", [1, 1, 1, 1])]),
             OW.node("HardSigmoid", ["c"], ["s"], "/act/HardSigmoid") if not as_mul else
             OW.node("Clip", ["c"], ["s"], "/act/Clip", [OW.attr_float("min", -1.0), OW.attr_float("max", 1.0)]),     # a Clip that is not ReLU6
             OW.node("Conv", ["s", "w2"], ["o1"], "/h1", [OW.attr_ints("kernel_shape", [1, 1])]),
             OW.node("Concat", ["o1", "o1"], ["cc"], "/cat", [OW.attr_int("axis", 1)]),
             OW.node("Reshape", ["cc", "shp"], ["r"], "/r")]
    inits = [OW.tensor("w", np.zeros((8, 3, 3, 3), np.float32)), OW.tensor("w2", np.zeros((64, 8, 1, 1), np.float32)), OW.tensor("shp", np.asarray([1, 128, -1], np.int64))]
    bad = tmp_path / "bad.onnx"
    open(bad, "wb").write(OW.model(nodes, inits, [("images", [1, 3, 32, 32])], [("r", [1, 128, 1024])]))
    with pytest.raises(ValueError, match="HardSigmoid|Clip|Detect head"):
        OL.lower(OI.read_onnx(str(bad)))


def test_efficientdet_d0_head_only_export_round_trip(tmp_path):
    """The whole EfficientDet-D0 graph (253 operators: MBConv + squeeze-and-excitation, three BiFPN cells, shared separable heads) written as a
    head-only ONNX export -- two outputs, box regression (1, A, 4) and class logits (1, A, 90) as axis-1 Concats of per-level
    Reshape(Transpose(conv)) -- and lowered back: the same op list, the ten per-level outputs EfficientdetEngine feeds to the device tail."""
    ws = M.SynthWeights(0, gain=M.synth_gain("efficientdet-d0"))
    g = M.build("efficientdet-d0", wsrc=ws, imgsz=(128, 128))
    path = str(tmp_path / "effdet_heads.onnx")
    onnx_emit.emit(g, path)
    m = OI.read_onnx(path)
    assert [o for o, _ in m.outputs] == ["regression", "classification"] and m.outputs[0][1] == [1, 9 * (256 + 64 + 16 + 4 + 1), 4]
    hipm, g2 = OI.convert(path, path + ".hipm")
    assert [nm for _, _, _, nm in g2.outs] == [nm for _, _, _, nm in g.outs] and len(g2.ops) == len(g.ops)
    x = np.random.default_rng(0).uniform(-1, 1, (2, 3, 128, 128)).astype(np.float32)
    a, b = graph_interp.run(g, x), graph_interp.run(g2, x)
    assert len(a) == len(b) == 10
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    kinds = [o["type"] for o in g2.ops]
    assert kinds.count(M.OP_SE_GATE) == 16 and kinds.count(M.OP_WSUM) == 24 and kinds.count(M.OP_UPSAMPLE2) == 0


def fuse_graph(hw=128):
    ws = M.SynthWeights(4, gain=1.0)
    g = M.Graph("fuse", 3, hw, hw, ws)
    x, cin = g.input()
    x = g.conv(x, 16, 3, 2, "stem", true_cin=cin)
    x = g.conv(x, 16, 3, 2, "c2")
    p3 = g.conv(x, 32, 3, 2, "p3")                                                             # stride 8 / 16 / 32 pyramid
    p4 = g.conv(p3, 32, 3, 2, "p4")
    p5 = g.conv(p4, 32, 3, 2, "p5")
    t = g.conv(p4, 32, 3, 1, "blk.cv1")
    p4 = g.conv(t, 32, 3, 1, "blk.cv2", res=p4, res_mode=M.RES_AFTER_ACT)                     # a residual: stays in the conv
    td4 = g.conv(g.wsum([p4, p5], M.fusion_weights([1.3, 0.6]), "td4.fuse"), 32, 3, 1, "td4.conv")          # weighted, upsampled term, swish
    td3 = g.wsum([p3, td4], [1.0, 1.0], "td3.sum", act=M.ACT_NONE)                                           # plain sum, upsampled term
    bu4 = g.conv(g.wsum([p4, td4, g.maxpool(td3, 3, 2, 1, name="ds")], M.fusion_weights([0.9, 1.1, 0.4]), "bu4.fuse"), 32, 3, 1, "bu4.conv")
    s5 = g.wsum([p5, g.maxpool(bu4, 3, 2, 1, name="ds2")], [1.0, 0.5], "s5.sum", act=M.ACT_RELU)              # ReLU behind the sum
    s5 = g.wsum([s5, p5], [1.0, -0.75], "l5.sum", act=M.ACT_LEAKY)                                            # LeakyReLU(0.1) behind the sum (round 5)
    ins, strides, nc = [], [], 7
    for i, f in enumerate((td3, bu4, s5)):
        ins += [g.conv(f, 64, 1, 1, "head.cv2.%d" % i, act=M.ACT_NONE, f32_out=True), g.conv(f, nc, 1, 1, "head.cv3.%d" % i, act=M.ACT_NONE, f32_out=True)]
        strides.append(hw // f.h)
    A = sum(f.h * f.w for f in (td3, bu4, s5))
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(M.OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    return g


def test_unsupported_nodes_fail_loudly_naming_the_node(tmp_path):
    """Nodes outside the recognised patterns (a bare Softmax that is not ultralytics-style attention) have no counterpart: the error names
    the node and the op."""
    w = np.zeros((8, 3, 3, 3), np.float32)
    nodes = [OW.node("Conv", ["images", "w"], ["c"], "/c", [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_ints("pads", [1, 1, 1, 1])]),
             OW.node("Softmax", ["c"], ["s"], "/attn/Softmax", [OW.attr_int("axis", 1)]),
             OW.node("Conv", ["s", "w2"], ["o1"], "/h1", [OW.attr_ints("kernel_shape", [1, 1])]),
             OW.node("Concat", ["o1", "o1"], ["cc"], "/cat", [OW.attr_int("axis", 1)]),
             OW.node("Reshape", ["cc", "shp"], ["r"], "/r")]
    inits = [OW.tensor("w", w), OW.tensor("w2", np.zeros((64, 8, 1, 1), np.float32)), OW.tensor("shp", np.asarray([1, 128, -1], np.int64))]
    p = tmp_path / "bad.onnx"
    open(p, "wb").write(OW.model(nodes, inits, [("images", [1, 3, 32, 32])], [("r", [1, 128, 1024])]))
    with pytest.raises(ValueError, match="Detect head|Softmax"):
        OL.lower(OI.read_onnx(str(p)))
    with pytest.raises(ValueError, match="neither a hand-built architecture"):
        OI.convert(str(p))


def test_sum_with_a_bilinear_resize_is_not_folded(tmp_path):
    """Only a nearest x2 Resize folds into a weighted sum: any other up-sampling in front of an Add stays its own node and is refused by name."""
    w = np.random.default_rng(0).standard_normal((16, 3, 3, 3)).astype(np.float32)
    w2 = np.random.default_rng(1).standard_normal((16, 16, 3, 3)).astype(np.float32)
    nodes = [OW.node("Conv", ["images", "w"], ["a"], "/a", [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_ints("pads", [1, 1, 1, 1])]),
             OW.node("Conv", ["a", "w2"], ["b"], "/b", [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_ints("pads", [1, 1, 1, 1]), OW.attr_ints("strides", [2, 2])]),
             OW.node("Resize", ["b", "", "sc"], ["u"], "/neck/Resize", [OW.attr_str("mode", "linear")]),
             OW.node("Add", ["a", "u"], ["s"], "/neck/Add"),
             OW.node("Conv", ["s", "wh"], ["o1"], "/h1", [OW.attr_ints("kernel_shape", [1, 1])]),
             OW.node("Conv", ["s", "wc"], ["o2"], "/h2", [OW.attr_ints("kernel_shape", [1, 1])]),
             OW.node("Concat", ["o1", "o2"], ["c0"], "/c0", [OW.attr_int("axis", 1)]), OW.node("Reshape", ["c0", "shp"], ["r0"], "/r0"),
             OW.node("Concat", ["r0", "r0", "r0"], ["out"], "/cat", [OW.attr_int("axis", 2)])]
    inits = [OW.tensor("w", w), OW.tensor("w2", w2), OW.tensor("sc", np.asarray([1, 1, 2, 2], np.float32)), OW.tensor("wh", np.zeros((64, 16, 1, 1), np.float32)),
             OW.tensor("wc", np.zeros((8, 16, 1, 1), np.float32)), OW.tensor("shp", np.asarray([1, 72, -1], np.int64))]
    p = tmp_path / "bilinear.onnx"
    open(p, "wb").write(OW.model(nodes, inits, [("images", [1, 3, 32, 32])], [("out", [1, 12, 3072])]))
    with pytest.raises(ValueError, match="neck/Resize|nearest"):
        OL.lower(OI.read_onnx(str(p)))


def test_nearest_x4_resize_becomes_a_chain_of_x2_launches(tmp_path):
    """YOLOv9 CBFuse style: a deeper map up-sampled x4 (nearest) and added to a shallow one.  The x4 Resize runs as two x2 launches, the Add as a
    weighted-sum op; values checked against a direct torch evaluation of the ONNX graph's meaning."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(2)
    w = (rng.standard_normal((16, 3, 3, 3)) * 0.2).astype(np.float32)
    w2 = (rng.standard_normal((16, 16, 3, 3)) * 0.1).astype(np.float32)
    w3 = (rng.standard_normal((16, 16, 3, 3)) * 0.1).astype(np.float32)
    wh = (rng.standard_normal((64, 16, 1, 1)) * 0.1).astype(np.float32)
    wc = (rng.standard_normal((8, 16, 1, 1)) * 0.1).astype(np.float32)
    k3 = [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_ints("pads", [1, 1, 1, 1])]
    w0 = (rng.standard_normal((16, 16, 3, 3)) * 0.1).astype(np.float32)
    nodes = [OW.node("Conv", ["images", "w"], ["p1"], "/p1", k3 + [OW.attr_ints("strides", [2, 2])]),
             OW.node("Conv", ["p1", "w0"], ["p2"], "/p2", k3 + [OW.attr_ints("strides", [2, 2])]),
             OW.node("Conv", ["p2", "w0"], ["a"], "/a", k3 + [OW.attr_ints("strides", [2, 2])]),          # stride 8
             OW.node("Conv", ["a", "w2"], ["b"], "/b", k3 + [OW.attr_ints("strides", [2, 2])]),
             OW.node("Conv", ["b", "w3"], ["c"], "/c", k3 + [OW.attr_ints("strides", [2, 2])]),
             OW.node("Resize", ["c", "", "sc"], ["u"], "/cbfuse/Resize", [OW.attr_str("mode", "nearest")]),
             OW.node("Add", ["a", "u"], ["s"], "/cbfuse/Add")]
    heads = []
    for l, src in enumerate(("s", "b", "c")):
        nodes += [OW.node("Conv", [src, "wh"], ["o1_%d" % l], "/h1_%d" % l, [OW.attr_ints("kernel_shape", [1, 1])]),
                  OW.node("Conv", [src, "wc"], ["o2_%d" % l], "/h2_%d" % l, [OW.attr_ints("kernel_shape", [1, 1])]),
                  OW.node("Concat", ["o1_%d" % l, "o2_%d" % l], ["c_%d" % l], "/c_%d" % l, [OW.attr_int("axis", 1)]),
                  OW.node("Reshape", ["c_%d" % l, "shp"], ["r_%d" % l], "/r_%d" % l)]
        heads.append("r_%d" % l)
    nodes.append(OW.node("Concat", heads, ["out"], "/cat", [OW.attr_int("axis", 2)]))
    inits = [OW.tensor("w", w), OW.tensor("w0", w0), OW.tensor("w2", w2), OW.tensor("w3", w3), OW.tensor("sc", np.asarray([1, 1, 4, 4], np.float32)), OW.tensor("wh", wh),
             OW.tensor("wc", wc), OW.tensor("shp", np.asarray([1, 72, -1], np.int64))]
    A = 16 * 16 + 8 * 8 + 4 * 4
    p = tmp_path / "x4.onnx"
    open(p, "wb").write(OW.model(nodes, inits, [("images", [1, 3, 128, 128])], [("out", [1, 12, A])]))
    g2 = OL.lower(OI.read_onnx(str(p)), "x4")
    assert sum(o["type"] == M.OP_UPSAMPLE2 for o in g2.ops) == 2 and sum(o["type"] == M.OP_WSUM for o in g2.ops) == 1
    x = rng.uniform(0, 1, (2, 3, 128, 128)).astype(np.float32)
    taps = {}
    graph_interp.run(g2, x, taps=taps)
    with torch.no_grad():
        t = torch.from_numpy(x)
        t = F.conv2d(F.conv2d(t, torch.from_numpy(w), stride=2, padding=1), torch.from_numpy(w0), stride=2, padding=1)
        a = F.conv2d(t, torch.from_numpy(w0), stride=2, padding=1)
        c = F.conv2d(F.conv2d(a, torch.from_numpy(w2), stride=2, padding=1), torch.from_numpy(w3), stride=2, padding=1)
        want = a + F.interpolate(c, scale_factor=4, mode="nearest")
    got = [v for k, v in taps.items() if "Add" in k or "sum" in k][0]
    np.testing.assert_allclose(got, want.numpy(), rtol=0, atol=1e-6)


def test_constant_nodes_are_read_like_initializers(tmp_path):
    """Exports without constant folding carry Reshape shapes / Resize scales / scalar factors as Constant nodes: the reader folds them into
    the initializer table, the lowering sees no difference."""
    g = fuse_graph()
    path = str(tmp_path / "fuse.onnx")
    onnx_emit.emit(g, path)
    m = OI.read_onnx(path)
    # rewrite every non-weight initializer (shapes, scales, scalar weights) as a Constant node in front of the graph
    small = {k: v for k, v in m.initializers.items() if np.asarray(v).size <= 8 and not k.endswith((".weight", ".bias"))}
    assert len(small) >= 8
    import onnx_writer as W2
    nodes = [W2.node("Constant", [], [k], "/Constant_%d" % i, [W2.attr_tensor("value", np.asarray(v))]) for i, (k, v) in enumerate(small.items())]
    for nd in m.nodes:
        attrs = []
        for k, v in nd["attrs"].items():
            if isinstance(v, (bytes, bytearray)):
                attrs.append(W2.attr_str(k, bytes(v).decode()))
            elif isinstance(v, list):
                attrs.append(W2.attr_ints(k, [int(x) for x in v]))
            elif isinstance(v, float):
                attrs.append(W2.attr_float(k, v))
            else:
                attrs.append(W2.attr_int(k, int(v)))
        nodes.append(W2.node(nd["op"], nd["inputs"], nd["outputs"], nd["name"], attrs))
    inits = [W2.tensor(k, np.asarray(v)) for k, v in m.initializers.items() if k not in small]
    p2 = str(tmp_path / "fuse_const.onnx")
    open(p2, "wb").write(W2.model(nodes, inits, m.inputs, m.outputs))
    m2 = OI.read_onnx(p2)
    assert not any(nd["op"] == "Constant" for nd in m2.nodes) and set(small) <= set(m2.initializers)
    g2 = OL.lower(m2, "c")
    x = np.random.default_rng(0).uniform(0, 1, (2, 3, g.in_h, g.in_w)).astype(np.float32)
    np.testing.assert_array_equal(graph_interp.run(g, x)[0], graph_interp.run(g2, x)[0])


def test_older_export_spellings_auto_pad_and_upsample_scales_attribute(tmp_path):
    """auto_pad = SAME_UPPER on stride-1 odd-kernel convolutions (= symmetric k // 2 padding) and the opset-7 Upsample node with its scales
    as an attribute mean what the explicit spellings mean; SAME_UPPER on a stride-2 convolution (asymmetric padding) is refused by name."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((16, 3, 3, 3)) * 0.2).astype(np.float32)
    w0 = (rng.standard_normal((16, 16, 3, 3)) * 0.1).astype(np.float32)
    wh, wc = (rng.standard_normal((64, 16, 1, 1)) * 0.1).astype(np.float32), (rng.standard_normal((8, 16, 1, 1)) * 0.1).astype(np.float32)
    s2 = [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_ints("pads", [1, 1, 1, 1]), OW.attr_ints("strides", [2, 2])]
    same = [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_str("auto_pad", "SAME_UPPER")]

    def graph(bad=False):
        nodes = [OW.node("Conv", ["images", "w"], ["p1"], "/p1", s2), OW.node("Conv", ["p1", "w0"], ["p2"], "/p2", s2), OW.node("Conv", ["p2", "w0"], ["a"], "/a", s2),
                 OW.node("Conv", ["a", "w0"], ["a2"], "/same", same),
                 OW.node("Conv", ["a2", "w0"], ["b"], "/b", (same + [OW.attr_ints("strides", [2, 2])]) if bad else s2),
                 OW.node("Upsample", ["b"], ["u"], "/up", [OW.attr_str("mode", "nearest"), OW.attr_floats("scales", [1.0, 1.0, 2.0, 2.0])]),
                 OW.node("Add", ["a2", "u"], ["s"], "/sum"),
                 OW.node("Conv", ["b", "w0"], ["c"], "/c", s2)]
        heads = []
        for l, src in enumerate(("s", "b", "c")):
            nodes += [OW.node("Conv", [src, "wh"], ["o1_%d" % l], "/h1_%d" % l, [OW.attr_ints("kernel_shape", [1, 1])]),
                      OW.node("Conv", [src, "wc"], ["o2_%d" % l], "/h2_%d" % l, [OW.attr_ints("kernel_shape", [1, 1])]),
                      OW.node("Concat", ["o1_%d" % l, "o2_%d" % l], ["c_%d" % l], "/c_%d" % l, [OW.attr_int("axis", 1)]),
                      OW.node("Reshape", ["c_%d" % l, "shp"], ["r_%d" % l], "/r_%d" % l)]
            heads.append("r_%d" % l)
        nodes.append(OW.node("Concat", heads, ["out"], "/cat", [OW.attr_int("axis", 2)]))
        inits = [OW.tensor("w", w), OW.tensor("w0", w0), OW.tensor("wh", wh), OW.tensor("wc", wc), OW.tensor("shp", np.asarray([1, 72, -1], np.int64))]
        return OW.model(nodes, inits, [("images", [1, 3, 128, 128])], [("out", [1, 12, 16 * 16 + 8 * 8 + 4 * 4])])

    p = tmp_path / "old.onnx"
    open(p, "wb").write(graph())
    g2 = OL.lower(OI.read_onnx(str(p)), "old")
    assert sum(o["type"] == M.OP_WSUM for o in g2.ops) == 1 and not any(o["type"] == M.OP_UPSAMPLE2 for o in g2.ops)      # the Upsample folded into the sum
    x = rng.uniform(0, 1, (2, 3, 128, 128)).astype(np.float32)
    taps = {}
    graph_interp.run(g2, x, taps=taps)
    with torch.no_grad():
        t = torch.from_numpy(x)
        for wt in (w, w0, w0):
            t = F.conv2d(t, torch.from_numpy(wt), stride=2, padding=1)
        a2 = F.conv2d(t, torch.from_numpy(w0), padding=1)
        b = F.conv2d(a2, torch.from_numpy(w0), stride=2, padding=1)
        want = a2 + F.interpolate(b, scale_factor=2, mode="nearest")
    np.testing.assert_allclose(taps["sum"], want.numpy(), rtol=0, atol=1e-6)
    open(p, "wb").write(graph(bad=True))
    with pytest.raises(ValueError, match="auto_pad SAME_UPPER"):
        OL.lower(OI.read_onnx(str(p)), "bad")


def test_unfused_batchnorm_is_folded_into_the_convolution(tmp_path):
    """Conv -> BatchNormalization -> ReLU (an export that kept its BatchNorm nodes): one convolution with folded weights, values = torch's."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(11)
    w = (rng.standard_normal((16, 3, 3, 3)) * 0.2).astype(np.float32)
    w0 = (rng.standard_normal((16, 16, 3, 3)) * 0.1).astype(np.float32)
    gam, bet = rng.uniform(0.5, 1.5, 16).astype(np.float32), rng.normal(0, 0.1, 16).astype(np.float32)
    mu, var = rng.normal(0, 0.2, 16).astype(np.float32), rng.uniform(0.5, 2.0, 16).astype(np.float32)
    wh, wc = (rng.standard_normal((64, 16, 1, 1)) * 0.1).astype(np.float32), (rng.standard_normal((8, 16, 1, 1)) * 0.1).astype(np.float32)
    s2 = [OW.attr_ints("kernel_shape", [3, 3]), OW.attr_ints("pads", [1, 1, 1, 1]), OW.attr_ints("strides", [2, 2])]
    nodes = [OW.node("Conv", ["images", "w"], ["p1"], "/p1", s2), OW.node("Conv", ["p1", "w0"], ["p2"], "/p2", s2),
             OW.node("Conv", ["p2", "w0"], ["a0"], "/a", s2),
             OW.node("BatchNormalization", ["a0", "gam", "bet", "mu", "var"], ["a1"], "/a/bn", [OW.attr_float("epsilon", 1e-3)]),
             OW.node("Relu", ["a1"], ["a"], "/a/relu"),
             OW.node("Conv", ["a", "w0"], ["b"], "/b", s2), OW.node("Conv", ["b", "w0"], ["c"], "/c", s2)]
    heads = []
    for l, src in enumerate(("a", "b", "c")):
        nodes += [OW.node("Conv", [src, "wh"], ["o1_%d" % l], "/h1_%d" % l, [OW.attr_ints("kernel_shape", [1, 1])]),
                  OW.node("Conv", [src, "wc"], ["o2_%d" % l], "/h2_%d" % l, [OW.attr_ints("kernel_shape", [1, 1])]),
                  OW.node("Concat", ["o1_%d" % l, "o2_%d" % l], ["c_%d" % l], "/c_%d" % l, [OW.attr_int("axis", 1)]),
                  OW.node("Reshape", ["c_%d" % l, "shp"], ["r_%d" % l], "/r_%d" % l)]
        heads.append("r_%d" % l)
    nodes.append(OW.node("Concat", heads, ["out"], "/cat", [OW.attr_int("axis", 2)]))
    inits = [OW.tensor("w", w), OW.tensor("w0", w0), OW.tensor("gam", gam), OW.tensor("bet", bet), OW.tensor("mu", mu), OW.tensor("var", var),
             OW.tensor("wh", wh), OW.tensor("wc", wc), OW.tensor("shp", np.asarray([1, 72, -1], np.int64))]
    p = tmp_path / "bn.onnx"
    open(p, "wb").write(OW.model(nodes, inits, [("images", [1, 3, 128, 128])], [("out", [1, 12, 336])]))
    g2 = OL.lower(OI.read_onnx(str(p)), "bn")
    x = rng.uniform(0, 1, (2, 3, 128, 128)).astype(np.float32)
    taps = {}
    graph_interp.run(g2, x, taps=taps)
    with torch.no_grad():
        t = torch.from_numpy(x)
        for wt in (w, w0, w0):
            t = F.conv2d(t, torch.from_numpy(wt), stride=2, padding=1)
        want = F.relu(F.batch_norm(t, torch.from_numpy(mu), torch.from_numpy(var), torch.from_numpy(gam), torch.from_numpy(bet), False, 0.0, 1e-3))
    name = [o["name"] for o in g2.ops if o["type"] == M.OP_CONV and o["act"] == M.ACT_RELU][0]
    np.testing.assert_allclose(taps[name], want.numpy(), rtol=0, atol=2e-6)
