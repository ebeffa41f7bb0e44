"""CPU: ONNX import (vehicle-cv-adas_amd/onnx_import.py) on files written by tests/onnx_writer.py in the conventions of the
reference's two exporters: ultralytics (fused, PyTorch parameter names kept) and torch.onnx.export of the eval-mode UFLDv2
model (BN folded by constant folding: anonymous `onnx::Conv_N` weights in execution order; BN kept: names kept)."""
import importlib, os
import numpy as np
import pytest

import onnx_writer as OW
from conftest import load_pkg

load_pkg()
M = importlib.import_module("adas_amd.models")
OI = importlib.import_module("adas_amd.onnx_import")
CE = importlib.import_module("adas_amd.coreEngine")


def synth(arch, **kw):
    ws = M.SynthWeights(3, gain=1.0)
    g = M.build(arch, wsrc=ws, **kw)
    return dict(ws.store), g


def conv_node(i, wname, bname, x, y):
    return OW.node("Conv", [x, wname] + ([bname] if bname else []), [y], "Conv_%d" % i, [OW.attr_ints("kernel_shape", [3, 3])])


def test_yolov8n_fused_names(tmp_path):
    W, g = synth("yolov8n")
    names = [k[:-7] for k in W if k.endswith(".weight")]
    inits, nodes = [], []
    for i, base in enumerate(names):                       # graph order = builder order; names carry the mapping
        inits.append(OW.tensor(base + ".weight", W[base + ".weight"], raw=(i % 2 == 0)))
        inits.append(OW.tensor(base + ".bias", W[base + ".bias"].astype(np.float32)))
        nodes.append(conv_node(i, base + ".weight", base + ".bias", "t%d" % i, "t%d" % (i + 1)))
    p = tmp_path / "yolov8n.onnx"
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("output0", [1, 84, 8400])]))
    m = OI.read_onnx(str(p))
    assert OI.detect_arch(m) == ("yolov8n", dict(nc=80, imgsz=(640, 640)))
    out, g2 = OI.convert(str(p), str(tmp_path / "y.hipm"))
    assert g2.tobytes() == M.build("yolov8n", wsrc=M.DictWeights(W)).tobytes()
    assert abs(g2.flops / 1e9 - 8.74) < 0.01


def test_yolov8s_unfused_bn_and_fp16(tmp_path):
    """Conv + BatchNorm kept separate under their PyTorch names, half-precision initializers: BN folded with eps 1e-3."""
    W, g = synth("yolov8s")
    rng = np.random.default_rng(0)
    inits, nodes, folded = [], [], {}
    for i, base in enumerate(k[:-7] for k in list(W) if k.endswith(".weight")):
        w, b = W[base + ".weight"], W[base + ".bias"]
        if base.endswith(".conv"):
            stem = base[:-5]
            c = w.shape[0]
            gmm, bt, mu, var = (rng.uniform(0.5, 1.5, c), rng.normal(0, .1, c), rng.normal(0, .1, c), rng.uniform(0.5, 1.5, c))
            gmm, bt, mu, var = (a.astype(np.float16).astype(np.float32) for a in (gmm, bt, mu, var))
            w16 = w.astype(np.float16)
            inits.append(OW.tensor(base + ".weight", w16))
            for suf, a in ((".weight", gmm), (".bias", bt), (".running_mean", mu), (".running_var", var)):
                inits.append(OW.tensor(stem + ".bn" + suf, a.astype(np.float16)))
            folded[base + ".weight"], folded[base + ".bias"] = OI.fold_bn(w16.astype(np.float32), None, gmm, bt, mu, var, 1e-3)
            nodes.append(conv_node(i, base + ".weight", None, "t%d" % i, "t%d" % (i + 1)))
        else:
            inits.append(OW.tensor(base + ".weight", w)); inits.append(OW.tensor(base + ".bias", b))
            folded[base + ".weight"], folded[base + ".bias"] = w, b
            nodes.append(conv_node(i, base + ".weight", base + ".bias", "t%d" % i, "t%d" % (i + 1)))
    p = tmp_path / "v8s.onnx"
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("output0", [1, 84, 8400])]))
    out, g2 = OI.convert(str(p), str(tmp_path / "s.hipm"))
    assert g2.name == "yolov8s"
    assert g2.tobytes() == M.build("yolov8s", wsrc=M.DictWeights(folded)).tobytes()


UFLD_KW = dict(in_h=160, in_w=800, num_grid_row=100, num_cls_row=36, num_grid_col=50, num_cls_col=41)


def torch_conv_order(depth="18"):
    names, cin = ["model.conv1"], 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], M.RESNET_DEPTHS[depth])):
        for bi in range(nblk):
            base = "model.layer%d.%d" % (li + 1, bi)
            names += [base + ".conv1", base + ".conv2"]
            if (li > 0 and bi == 0) or cin != planes:
                names.append(base + ".downsample.0")
            cin = planes
    return names + ["pool"]


def test_ufldv2_anonymous_convs_in_execution_order(tmp_path):
    """torch.onnx.export of the eval-mode model: BN constant-folded into `onnx::Conv_N` tensors (names gone), Linear layers
    as Gemm(transB=1) under their own names, LayerNorm affine under its own name."""
    W, g = synth("ufldv2_res18", **UFLD_KW)
    inits, nodes = [], []
    for i, base in enumerate(torch_conv_order()):
        wn, bn = ("onnx::Conv_%d" % (200 + 2 * i), "onnx::Conv_%d" % (201 + 2 * i)) if base != "pool" else ("pool.weight", "pool.bias")
        inits += [OW.tensor(wn, W[base + ".weight"]), OW.tensor(bn, W[base + ".bias"])]
        nodes.append(conv_node(i, wn, bn, "t%d" % i, "t%d" % (i + 1)))
    for nm in ("cls.0.weight", "cls.0.bias", "cls.1.bias", "cls.3.bias"):
        inits.append(OW.tensor(nm, W[nm]))
    inits.append(OW.tensor("cls.1.weight", W["cls.1.weight"]))
    nodes.append(OW.node("Gemm", ["f", "cls.1.weight", "cls.1.bias"], ["h"], "Gemm_0", [OW.attr_int("transB", 1)]))
    inits.append(OW.tensor("onnx::MatMul_900", np.ascontiguousarray(W["cls.3.weight"].T)))       # (in, out), name lost
    nodes.append(OW.node("MatMul", ["h", "onnx::MatMul_900"], ["o"], "MatMul_0"))
    outs = [("loc_row", [1, 100, 36, 4]), ("loc_col", [1, 50, 41, 4]), ("exist_row", [1, 2, 36, 4]), ("exist_col", [1, 2, 41, 4])]
    p = tmp_path / "culane_res18.onnx"
    p.write_bytes(OW.model(nodes, inits, [("input", [1, 3, 160, 800])], outs))
    m = OI.read_onnx(str(p))
    arch, kw = OI.detect_arch(m)
    assert arch == "ufldv2_res18" and kw["num_grid_row"] == 100 and kw["in_w"] == 800
    out, g2 = OI.convert(str(p), str(tmp_path / "l.hipm"))
    assert g2.tobytes() == M.build("ufldv2_res18", wsrc=M.DictWeights(W), **UFLD_KW).tobytes()


def test_ufldv2_named_convs_with_batchnorm_nodes(tmp_path):
    """Export without constant folding: Conv (no bias) + BatchNormalization nodes, PyTorch names kept: BN folded (eps 1e-5)."""
    W, g = synth("ufldv2_res18", **UFLD_KW)
    rng = np.random.default_rng(1)
    inits, nodes, folded = [], [], dict(W)
    for i, base in enumerate(torch_conv_order()):
        w, b = W[base + ".weight"], W[base + ".bias"]
        if base == "pool":
            inits += [OW.tensor("pool.weight", w), OW.tensor("pool.bias", b)]
            nodes.append(conv_node(i, "pool.weight", "pool.bias", "t%d" % i, "t%d" % (i + 1)))
            continue
        bn = base.replace(".downsample.0", ".downsample.1") if "downsample" in base else base[:-5] + "bn" + base[-1]
        c = w.shape[0]
        gmm, bt, mu, var = (rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, .1, c).astype(np.float32),
                            rng.normal(0, .1, c).astype(np.float32), rng.uniform(0.5, 1.5, c).astype(np.float32))
        inits.append(OW.tensor(base + ".weight", w))
        for suf, a in ((".weight", gmm), (".bias", bt), (".running_mean", mu), (".running_var", var)):
            inits.append(OW.tensor(bn + suf, a))
        nodes.append(conv_node(i, base + ".weight", None, "t%d" % i, "c%d" % i))
        nodes.append(OW.node("BatchNormalization", ["c%d" % i] + [bn + s for s in (".weight", ".bias", ".running_mean", ".running_var")],
                             ["t%d" % (i + 1)], "BN_%d" % i, [OW.attr_float("epsilon", 1e-5)]))
        folded[base + ".weight"], folded[base + ".bias"] = OI.fold_bn(w, None, gmm, bt, mu, var, 1e-5)
    for nm in ("cls.0.weight", "cls.0.bias", "cls.1.weight", "cls.1.bias", "cls.3.weight", "cls.3.bias"):
        inits.append(OW.tensor(nm, W[nm]))
    outs = [("loc_row", [1, 100, 36, 4]), ("loc_col", [1, 50, 41, 4]), ("exist_row", [1, 2, 36, 4]), ("exist_col", [1, 2, 41, 4])]
    p = tmp_path / "named.onnx"
    p.write_bytes(OW.model(nodes, inits, [("input", [1, 3, 160, 800])], outs))
    out, g2 = OI.convert(str(p), str(tmp_path / "n.hipm"))
    ref = M.build("ufldv2_res18", wsrc=M.DictWeights(folded), **UFLD_KW)
    assert g2.tobytes() == ref.tobytes()


def test_unsupported_and_garbage_fail_loudly(tmp_path):
    bad = tmp_path / "x.onnx"
    bad.write_bytes(b"\x08\x07not a container")
    with pytest.raises(ValueError):
        OI.read_onnx(str(bad))
    w = np.zeros((24, 3, 5, 5), np.float32)
    p = tmp_path / "other.onnx"
    p.write_bytes(OW.model([conv_node(0, "w", None, "x", "y")], [OW.tensor("w", w)], [("x", [1, 3, 224, 224])], [("y", [1, 1000])]))
    with pytest.raises(ValueError, match="not a supported architecture"):
        OI.convert(str(p))
    # HipEngine hands a non-ONNX '.onnx' file to the library untouched (it reports ADAS_ERR_FORMAT on a GPU box)
    assert CE.HipEngine._resolve_container(str(bad)) == str(bad)


def test_ufldv2_tusimple_detected_without_layernorm(tmp_path):
    kw = dict(in_h=64, in_w=160, num_grid_row=20, num_cls_row=8, num_grid_col=20, num_cls_col=6, fc_norm=False)
    W, g = synth("ufldv2_res18", **kw)
    assert "cls.0.weight" not in W
    inits, nodes = [], []
    for i, base in enumerate(torch_conv_order()):
        inits += [OW.tensor(base + ".weight", W[base + ".weight"]), OW.tensor(base + ".bias", W[base + ".bias"])]
        nodes.append(conv_node(i, base + ".weight", base + ".bias", "t%d" % i, "t%d" % (i + 1)))
    for nm in ("cls.1.weight", "cls.1.bias", "cls.3.weight", "cls.3.bias"):
        inits.append(OW.tensor(nm, W[nm]))
    outs = [("loc_row", [1, 20, 8, 4]), ("loc_col", [1, 20, 6, 4]), ("exist_row", [1, 2, 8, 4]), ("exist_col", [1, 2, 6, 4])]
    p = tmp_path / "tusimple_res18.onnx"
    p.write_bytes(OW.model(nodes, inits, [("input", [1, 3, 64, 160])], outs))
    arch, got_kw = OI.detect_arch(OI.read_onnx(str(p)))
    assert arch == "ufldv2_res18" and got_kw["fc_norm"] is False
    out, g2 = OI.convert(str(p), str(tmp_path / "t.hipm"))
    assert g2.tobytes() == M.build("ufldv2_res18", wsrc=M.DictWeights(W), **kw).tobytes()


def test_ufld_v1_single_output_detected(tmp_path):
    """UFLD v1 export (tusimple_18.onnx style): ResNet stem, one (1, G+1, K, 4) output, Linear layers cls.0 / cls.2 as Gemm."""
    kw = dict(in_h=96, in_w=160, griding_num=20, cls_num_per_lane=8)
    W, g = synth("ufld_v1_res18", **kw)
    inits, nodes = [], []
    for i, base in enumerate(torch_conv_order()):
        wn, bn = ("onnx::Conv_%d" % (300 + 2 * i), "onnx::Conv_%d" % (301 + 2 * i)) if base != "pool" else ("pool.weight", "pool.bias")
        inits += [OW.tensor(wn, W[base + ".weight"]), OW.tensor(bn, W[base + ".bias"])]
        nodes.append(conv_node(i, wn, bn, "t%d" % i, "t%d" % (i + 1)))
    for nm in ("cls.0.weight", "cls.0.bias", "cls.2.weight", "cls.2.bias"):
        inits.append(OW.tensor(nm, W[nm]))
    nodes.append(OW.node("Gemm", ["f", "cls.0.weight", "cls.0.bias"], ["h"], "Gemm_0", [OW.attr_int("transB", 1)]))
    nodes.append(OW.node("Gemm", ["h", "cls.2.weight", "cls.2.bias"], ["o"], "Gemm_1", [OW.attr_int("transB", 1)]))
    p = tmp_path / "tusimple_18.onnx"
    p.write_bytes(OW.model(nodes, inits, [("input", [1, 3, 96, 160])], [("output", [1, 21, 8, 4])]))
    arch, got = OI.detect_arch(OI.read_onnx(str(p)))
    assert arch == "ufld_v1_res18" and got == dict(in_h=96, in_w=160, griding_num=20, cls_num_per_lane=8, num_lanes=4)
    out, g2 = OI.convert(str(p), str(tmp_path / "u1.hipm"))
    assert g2.tobytes() == M.build("ufld_v1_res18", wsrc=M.DictWeights(W), **kw).tobytes()


def test_fp16_export_and_input_size_checks(tmp_path):
    """An fp16 export (float16 graph input, onnxQuantization.py:11-41) is flagged in the container header (HipEngine.engine_dtype
    follows it, coreEngine.py:168); non-square inputs reach the builder as (H, W); dynamic or non-multiple-of-32 sizes fail loudly."""
    import struct
    W = {}
    g0 = M.build("yolov8n", wsrc=M.SynthWeights(0, gain=M.SILU_GAIN))
    ws = M.SynthWeights(0, gain=M.SILU_GAIN)
    M.build("yolov8n", wsrc=ws)
    inits, nodes = [], []
    for i, base in enumerate(k[:-7] for k in ws.store if k.endswith(".weight")):
        inits += [OW.tensor(base + ".weight", ws.store[base + ".weight"].astype(np.float16)),
                  OW.tensor(base + ".bias", ws.store[base + ".bias"].astype(np.float16))]
        nodes.append(OW.node("Conv", ["t%d" % i, base + ".weight", base + ".bias"], ["t%d" % (i + 1)], "Conv_%d" % i))
    p = tmp_path / "half.onnx"
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("output0", [1, 84, 8400])], elem_type=10))
    m = OI.read_onnx(str(p))
    assert m.elem_types["images"] == 10
    out, g = OI.convert(str(p), str(tmp_path / "half.hipm"))
    assert g.io_half
    hdr = struct.unpack(M.HDR_FMT, open(out, "rb").read(M.HDR_SIZE))
    assert hdr[8] == (8 | (1 << 16))                      # in_cpad word: bit 16 = float16 model I/O
    assert struct.unpack(M.HDR_FMT, g0.tobytes()[:M.HDR_SIZE])[8] == 8
    # an fp32 file converted with io_half=True: the onnxQuantization.py counterpart
    f32 = tmp_path / "full.onnx"
    inits32 = []
    for base in (k[:-7] for k in ws.store if k.endswith(".weight")):
        inits32 += [OW.tensor(base + ".weight", ws.store[base + ".weight"]), OW.tensor(base + ".bias", ws.store[base + ".bias"])]
    f32.write_bytes(OW.model(nodes, inits32, [("images", [1, 3, 640, 640])], [("output0", [1, 84, 8400])]))
    _, g32 = OI.convert(str(f32), str(tmp_path / "full.hipm"))
    _, g16 = OI.convert(str(f32), str(tmp_path / "full_fp16.hipm"), io_half=True)
    assert not g32.io_half and g16.io_half
    # non-square
    q = tmp_path / "rect.onnx"
    A = 48 * 80 + 24 * 40 + 12 * 20
    q.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 384, 640])], [("output0", [1, 84, A])]))
    assert OI.detect_arch(OI.read_onnx(str(q))) == ("yolov8n", dict(nc=80, imgsz=(384, 640)))
    gq = M.build("yolov8n", imgsz=(384, 640))
    assert (gq.in_h, gq.in_w) == (384, 640) and gq.meta["anchors"] == A
    for bad in ([1, 3, -1, -1], [1, 3, 0, 640], [1, 3, 600, 600]):
        r = tmp_path / "bad.onnx"
        r.write_bytes(OW.model(nodes, inits, [("images", bad)], [("output0", [1, 84, 8400])]))
        with pytest.raises(ValueError):
            OI.detect_arch(OI.read_onnx(str(r)))


def test_yolov10n_names_depthwise_and_unfused_repvggdw(tmp_path):
    """The reference's shipped default (demo.py:24-30) exported with the v8-layout head it decodes (yoloDetector.py:114,121):
    recognised by its PSA / one-to-one-head parameter names (or its depth-wise convolutions), weights taken by name, and an
    un-fused RepVGGDW (7x7 + 3x3 branches, each with its own BatchNorm) re-parameterised to the single 7x7 the graph runs."""
    W, g = synth("yolov10n")
    rng = np.random.default_rng(1)
    inits, nodes, want = [], [], dict(W)
    rep = "model.22.m.0.cv1.2"
    for i, base in enumerate(k[:-7] for k in list(W) if k.endswith(".weight")):
        w, b = W[base + ".weight"], W[base + ".bias"]
        grp = [OW.attr_int("group", w.shape[0])] if w.shape[1] == 1 and w.shape[0] > 1 else []
        if base == rep + ".conv.conv":
            # split the fused 7x7 into the two un-fused branches with BatchNorms: 7x7 part + 3x3 part
            c = w.shape[0]
            w3 = (rng.standard_normal((c, 1, 3, 3)) * 0.1).astype(np.float32)
            stats = {}
            for br in ("conv", "conv1"):
                gmm, bt, mu, var = (rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, .1, c).astype(np.float32),
                                    rng.normal(0, .1, c).astype(np.float32), rng.uniform(0.5, 1.5, c).astype(np.float32))
                stats[br] = (gmm, bt, mu, var)
                for suf, a in ((".weight", gmm), (".bias", bt), (".running_mean", mu), (".running_var", var)):
                    inits.append(OW.tensor("%s.%s.bn%s" % (rep, br, suf), a))
            w7 = (rng.standard_normal((c, 1, 7, 7)) * 0.1).astype(np.float32)
            inits.append(OW.tensor(rep + ".conv.conv.weight", w7))
            inits.append(OW.tensor(rep + ".conv1.conv.weight", w3))
            f7 = OI.fold_bn(w7, None, *stats["conv"], 1e-3)
            f3 = OI.fold_bn(w3, None, *stats["conv1"], 1e-3)
            want[base + ".weight"] = (f7[0] + np.pad(f3[0], ((0, 0), (0, 0), (2, 2), (2, 2)))).astype(np.float32)
            want[base + ".bias"] = (f7[1] + f3[1]).astype(np.float32)
            nodes.append(OW.node("Conv", ["t%d" % i, rep + ".conv.conv.weight"], ["t%da" % i], "Conv_%da" % i, [OW.attr_ints("kernel_shape", [7, 7])] + grp))
            nodes.append(OW.node("Conv", ["t%d" % i, rep + ".conv1.conv.weight"], ["t%db" % i], "Conv_%db" % i, [OW.attr_ints("kernel_shape", [3, 3])] + grp))
            continue
        inits.append(OW.tensor(base + ".weight", w)); inits.append(OW.tensor(base + ".bias", b))
        nodes.append(OW.node("Conv", ["t%d" % i, base + ".weight", base + ".bias"], ["t%d" % (i + 1)], "Conv_%d" % i,
                             [OW.attr_ints("kernel_shape", list(w.shape[2:]))] + grp))
    p = tmp_path / "yolov10n.onnx"
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("output0", [1, 84, 8400])]))
    m = OI.read_onnx(str(p))
    assert OI.detect_arch(m) == ("yolov10n", dict(nc=80, imgsz=(640, 640)))
    out, g2 = OI.convert(str(p), str(tmp_path / "v10.hipm"))
    assert g2.name == "yolov10n" and abs(g2.flops / 1e9 - 6.76) < 0.02 and abs(g2.n_params / 1e6 - 2.30) < 0.01
    ref = M.build("yolov10n", wsrc=M.DictWeights(want))
    assert g2.tobytes() == ref.tobytes()


@pytest.mark.parametrize("name", ["yolov9t", "yolov9s", "yolov9c"])
def test_yolov9t_recognised_by_average_pool_or_names(tmp_path, name):
    """YOLOv9t / s (GELAN) share YOLOv8n / s's stem width and (1, 84, 8400) head: told apart by their AConv average-pool nodes or
    RepNCSPELAN4 parameter names; weights by name; RepConv is expected in its fused (deploy) form, as ultralytics exports it."""
    W, g = synth(name)
    inits, nodes = [], []
    for i, base in enumerate(k[:-7] for k in list(W) if k.endswith(".weight")):
        w, b = W[base + ".weight"], W[base + ".bias"]
        inits.append(OW.tensor(base + ".weight", w)); inits.append(OW.tensor(base + ".bias", b))
        nodes.append(OW.node("Conv", ["t%d" % i, base + ".weight", base + ".bias"], ["t%d" % (i + 1)], "Conv_%d" % i,
                             [OW.attr_ints("kernel_shape", list(w.shape[2:]))]))
    nodes.insert(3, OW.node("AveragePool", ["t3"], ["t3p"], "AveragePool_0", [OW.attr_ints("kernel_shape", [2, 2])]))
    p = tmp_path / (name + ".onnx")
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("output0", [1, 84, 8400])]))
    m = OI.read_onnx(str(p))
    assert OI.detect_arch(m) == (name, dict(nc=80, imgsz=(640, 640)))
    out, g2 = OI.convert(str(p), str(tmp_path / "v9.hipm"))
    assert g2.name == name and g2.tobytes() == M.build(name, wsrc=M.DictWeights(W)).tobytes()


def test_yolov7_tiny_recognised_by_stem_and_v5_layout(tmp_path):
    """YOLOv7-tiny: (1, 25200, 85) head like YOLOv5 but a 3x3 32-channel stem and 58 convolutions (LeakyRelu nodes between them);
    weights by their upstream names (model.<row>.conv, model.77.m.<level>)."""
    W, g = synth("yolov7-tiny")
    inits, nodes = [], []
    for i, base in enumerate(k[:-7] for k in list(W) if k.endswith(".weight")):
        w, b = W[base + ".weight"], W[base + ".bias"]
        inits.append(OW.tensor(base + ".weight", w)); inits.append(OW.tensor(base + ".bias", b))
        nodes.append(OW.node("Conv", ["t%d" % i, base + ".weight", base + ".bias"], ["c%d" % i], "Conv_%d" % i,
                             [OW.attr_ints("kernel_shape", list(w.shape[2:]))]))
        nodes.append(OW.node("LeakyRelu", ["c%d" % i], ["t%d" % (i + 1)], "LeakyRelu_%d" % i, [OW.attr_float("alpha", 0.1)]))
    p = tmp_path / "yolov7-tiny.onnx"
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("output", [1, 25200, 85])]))
    m = OI.read_onnx(str(p))
    assert OI.detect_arch(m) == ("yolov7-tiny", dict(nc=80, imgsz=(640, 640)))
    out, g2 = OI.convert(str(p), str(tmp_path / "v7.hipm"))
    assert g2.name == "yolov7-tiny" and g2.tobytes() == M.build("yolov7-tiny", wsrc=M.DictWeights(W)).tobytes()
    q = tmp_path / "yolov7.onnx"                             # another YOLOv7 scale: refused by name, not mis-built
    q.write_bytes(OW.model(nodes[:40], inits, [("images", [1, 3, 640, 640])], [("output", [1, 25200, 85])]))
    with pytest.raises(ValueError, match="yolov7-tiny"):
        OI.detect_arch(OI.read_onnx(str(q)))


def test_yolov6n_weights_by_execution_order(tmp_path):
    """YOLOv6 v3.0 deploy export: anonymous initializers (the exporter drops upstream's module paths), 69 Conv + 2 ConvTranspose nodes in
    execution order, ConvTranspose weights in (Cin, Cout, 2, 2) layout; recognised by its transposed convs and (1, 8400, 85) output."""
    W, g = synth("yolov6n")
    inits, nodes = [], []
    for i, base in enumerate(k[:-7] for k in list(W) if k.endswith(".weight")):
        w, b = W[base + ".weight"], W[base + ".bias"]
        wn, bn = "onnx::Conv_%d" % (900 + 2 * i), "onnx::Conv_%d" % (901 + 2 * i)
        inits.append(OW.tensor(wn, w)); inits.append(OW.tensor(bn, b))
        op = "ConvTranspose" if "upsample_transpose" in base else "Conv"
        nodes.append(OW.node(op, ["t%d" % i, wn, bn], ["t%d" % (i + 1)], "%s_%d" % (op, i), [OW.attr_ints("kernel_shape", list(w.shape[2:]))]))
    p = tmp_path / "yolov6n.onnx"
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("outputs", [1, 8400, 85])]))
    m = OI.read_onnx(str(p))
    assert OI.detect_arch(m) == ("yolov6n", dict(nc=80, imgsz=(640, 640)))
    out, g2 = OI.convert(str(p), str(tmp_path / "v6.hipm"))
    assert g2.name == "yolov6n" and g2.tobytes() == M.build("yolov6n", wsrc=M.DictWeights(W)).tobytes()
