"""GPU parity of the network forward passes (HIP MFMA conv engine) against the torch-CPU fp32 oracle.

Tolerances (stated per BASELINE.json north_star: "within 1e-3 on conv activations"):
  * precision="fp32" (fp32 storage + fp32 MFMA, the parity mode): max |diff| <= 1e-3 on every tapped
    activation and on the head outputs.
  * precision="bf16" (bench mode): bf16 storage cannot meet an absolute 1e-3 on O(1..10) activations
    (half-ulp at 1.0 is 3.9e-3).  End to end through ~25 random-weight layers the rounding noise
    accumulates to a few % (measured 3.6e-2 rel-L2 at P3 with either conv kernel); bound: 6e-2.
    The kernels themselves are pinned per layer in tests/test_gpu_conv.py (rel-L2 <= 1e-2 over two
    bf16 layers, 1e-5 in fp32 mode).
"""
import numpy as np
import pytest

import netutil
from oracle import nets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def CE():
    import importlib
    from conftest import load_pkg
    load_pkg()
    ce = importlib.import_module("adas_amd.coreEngine")
    assert ce.L.lib().adas_device_count() > 0
    return ce


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30))


def test_engine_surface_and_errors(CE, tmp_path):
    with pytest.raises(Exception, match="can't not found"):
        CE.HipEngine(str(tmp_path / "missing.onnx"))
    bad = tmp_path / "real.onnx"
    bad.write_bytes(b"\x08\x07not a container")
    with pytest.raises(Exception, match="ADASHIP1"):
        CE.OnnxEngine(str(bad))
    path, W, g = netutil.model("yolov8n")
    e = CE.TensorRTEngine(path, precision="fp32")
    assert e.get_engine_input_shape() == [1, 3, 640, 640]
    shapes, names = e.get_engine_output_shape()
    assert shapes == [[1, 84, 8400]] and names == ["output0"]
    assert e.framework_type == "hip" and e.engine_dtype == np.float32
    st = e.stats()
    assert abs(st["flops_per_frame"] / 1e9 - 8.74) < 0.01
    e.close()


@pytest.mark.parametrize("prec,tol_abs,tol_rel", [("fp32", 1e-3, 1e-4), ("bf16", None, 6e-2)])
def test_yolov8n_vs_oracle(CE, prec, tol_abs, tol_rel):
    path, W, g = netutil.model("yolov8n")
    x = netutil.coco_like_frames(2)
    taps = {}
    want = nets.yolov8_forward(x, W, "n", taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    got = e.engine_inference(x)[0]
    assert got.shape == want.shape == (2, 84, 8400)
    # intermediate activations (conv stacks)
    for lname, key in (("model.15.cv2.conv", "p3"), ("model.18.cv2.conv", "p4"), ("model.21.cv2.conv", "p5"),
                       ("model.9.cv2.conv", "sppf")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err = np.abs(a - ref).max()
        print(prec, lname, "max|diff| %.3e  rel_l2 %.3e  max|ref| %.2f" % (err, rel_l2(a, ref), np.abs(ref).max()))
        if tol_abs is not None:
            assert err <= tol_abs, (lname, err)
        assert rel_l2(a, ref) <= tol_rel, lname
    box_err = np.abs(got[:, :4] - want[:, :4]).max()
    cls_err = np.abs(got[:, 4:] - want[:, 4:]).max()
    print(prec, "head: box max|diff| %.3e px, cls max|diff| %.3e" % (box_err, cls_err))
    if tol_abs is not None:
        assert cls_err <= tol_abs and box_err <= 1e-3 * max(1.0, float(np.abs(want[:, :4]).max()))   # pixels
    else:
        assert rel_l2(got, want) <= tol_rel
    e.close()


def test_yolov5n_plumbing_config_c1(CE):
    """BASELINE config 1 shape: YOLOv5n 640x640 single frame through the coreEngine surface."""
    path, W, g = netutil.model("yolov5n")
    x = netutil.coco_like_frames(1, seed=0)
    want = nets.yolov5_forward(x, W, "n")
    e = CE.OnnxEngine(path, precision="fp32")
    got = e.engine_inference(x)[0]
    assert got.shape == want.shape == (1, 25200, 85)
    err = np.abs(got[..., 4:] - want[..., 4:]).max()
    berr = np.abs(got[..., :4] - want[..., :4]).max()
    print("v5n fp32: score max|diff| %.3e, box max|diff| %.3e px" % (err, berr))
    # boxes are in input pixels (up to (2*sigmoid)^2 * 373 px): bound relative to their magnitude
    assert err <= 1e-3 and berr <= 1e-3 * max(1.0, float(np.abs(want[..., :4]).max()))
    e.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ufldv2_small_vs_oracle(CE, prec):
    """Reduced geometry (160x800 input, 100/50 grid cells, 36/41 anchors) -- same layer types, seconds to run."""
    kw = dict(in_h=160, in_w=800, num_grid_row=100, num_cls_row=36, num_grid_col=50, num_cls_col=41)
    path, W, g = netutil.model("ufldv2_res18", **kw)
    x = netutil.lane_frames(2, 160, 800)
    taps = {}
    want = nets.ufldv2_forward(x, W, "18", 100, 36, 50, 41, taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    got = e.engine_inference(x)
    assert [o.shape for o in got] == [w.shape for w in want]
    a = e.fetch_activation("model.layer4.1.conv2", 2)
    ref = taps["layer4"].numpy()
    print(prec, "layer4 max|diff| %.3e rel %.3e max|ref| %.2f" % (np.abs(a - ref).max(), rel_l2(a, ref), np.abs(ref).max()))
    for o, w, nm in zip(got, want, ("loc_row", "loc_col", "exist_row", "exist_col")):
        print(prec, nm, "max|diff| %.3e rel %.3e" % (np.abs(o - w).max(), rel_l2(o, w)))
        if prec == "fp32":
            assert np.abs(o - w).max() <= 1e-3
        else:
            assert rel_l2(o, w) <= 6e-2
    e.close()


def test_hipengine_loads_a_real_onnx_file(CE, tmp_path):
    """OnnxEngine('model.onnx') as the reference calls it (coreEngine.py:161-170): the ONNX file is converted once
    (onnx_import) and runs; outputs equal those of the directly built container."""
    import importlib
    import onnx_writer as OW
    OI = importlib.import_module("adas_amd.onnx_import")
    path, W, g = netutil.model("yolov8n")
    inits, nodes = [], []
    for i, base in enumerate(k[:-7] for k in W if k.endswith(".weight")):
        inits += [OW.tensor(base + ".weight", W[base + ".weight"]), OW.tensor(base + ".bias", W[base + ".bias"])]
        nodes.append(OW.node("Conv", ["t%d" % i, base + ".weight", base + ".bias"], ["t%d" % (i + 1)], "Conv_%d" % i))
    p = tmp_path / "yolov8n-coco.onnx"
    p.write_bytes(OW.model(nodes, inits, [("images", [1, 3, 640, 640])], [("output0", [1, 84, 8400])]))
    x = netutil.coco_like_frames(1)
    e1 = CE.OnnxEngine(str(p), precision="fp32")
    e2 = CE.HipEngine(path, precision="fp32")
    assert e1.get_engine_output_shape() == e2.get_engine_output_shape()
    np.testing.assert_array_equal(e1.engine_inference(x)[0], e2.engine_inference(x)[0])
    assert any(f.endswith(".hipm") for f in __import__("os").listdir(tmp_path))      # cached conversion next to the .onnx
    e1.close(); e2.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ufldv2_tusimple_variant_vs_oracle(CE, prec):
    """Tusimple configuration (configs/tusimple_res18.py: 800x320, 100/100 cells, 56/41 anchors, fc_norm=False): the 1x1 `pool`
    conv output is re-viewed as the flat FC input through a buffer alias instead of going through LayerNorm."""
    path, W, g = netutil.model("ufldv2_tusimple_res18")
    assert g.meta["fc_norm"] is False and g.in_h == 320 and g.in_w == 800
    x = netutil.lane_frames(3, 320, 800)
    want = nets.ufldv2_forward(x, W, "18", 100, 56, 100, 41, fc_norm=False)
    e = CE.HipEngine(path, precision=prec, max_batch=3)
    shapes, names = e.get_engine_output_shape()
    assert shapes == [[1, 100, 56, 4], [1, 100, 41, 4], [1, 2, 56, 4], [1, 2, 41, 4]] and names == ["loc_row", "loc_col", "exist_row", "exist_col"]
    got = e.engine_inference(x)
    for o, w, nm in zip(got, want, names):
        print(prec, nm, "max|diff| %.3e rel %.3e" % (np.abs(o - w).max(), rel_l2(o, w)))
        if prec == "fp32":
            assert np.abs(o - w).max() <= 1e-3
        else:
            assert rel_l2(o, w) <= 6e-2
    e.close()


@pytest.mark.parametrize("name,prec,G,K", [("ufld_v1_res18", "fp32", 100, 56), ("ufld_v1_res18", "bf16", 100, 56),
                                           ("ufld_v1_culane_res18", "fp32", 200, 18)])
def test_ufld_v1_vs_oracle(CE, name, prec, G, K):
    """UFLD v1 network (800x288 -> one (1, G+1, K, 4) tensor, ultrafastLaneDetector.py:73-75,99) vs the torch fp32 restatement."""
    path, W, g = netutil.model(name)
    assert g.in_h == 288 and g.in_w == 800
    x = netutil.lane_frames(2, 288, 800)
    want = nets.ufld_v1_forward(x, W, "18", G, K)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    shapes, names = e.get_engine_output_shape()
    assert shapes == [[1, G + 1, K, 4]] and len(names) == 1
    got = e.engine_inference(x)[0]
    print(name, prec, "max|diff| %.3e rel %.3e" % (np.abs(got - want).max(), rel_l2(got, want)))
    if prec == "fp32":
        assert np.abs(got - want).max() <= 1e-3
    else:
        assert rel_l2(got, want) <= 6e-2
    e.close()


@pytest.mark.parametrize("name,scale", [("yolov8m", "m"), ("yolov8x", "x"), ("yolov5m", "m"), ("yolov5x", "x")])
def test_wider_yolo_scales_vs_oracle(CE, name, scale):
    """The m/x width and depth multiples (README model table: yolov5n/s/m/l/x, yolov8n/s/m/l/x) at 256x256: channel counts
    that are not powers of two (48, 80, 96, 160, 320, ...) through the same kernels.  Tolerances: relative L2 of the head
    tensor <= 2e-4 in fp32 and <= 8e-2 in bf16.  No absolute bound on the class probabilities here: with random weights the
    logits of these 2-4x deeper nets grow to thousands, so a 1e-6 relative summation-order difference against oneDNN moves
    a near-zero logit by 1e-2 and its sigmoid visibly (the n-scale tests keep the absolute 1e-3 bound)."""
    path, W, g = netutil.model(name, imgsz=256)
    x = netutil.coco_like_frames(2, 256, 256)
    v8 = name.startswith("yolov8")
    want = nets.yolov8_forward(x, W, scale) if v8 else nets.yolov5_forward(x, W, scale)
    for prec in ("fp32", "bf16"):
        e = CE.HipEngine(path, precision=prec, max_batch=2)
        got = e.engine_inference(x)[0]
        assert got.shape == want.shape
        cls_g, cls_w = (got[:, 4:], want[:, 4:]) if v8 else (got[..., 4:], want[..., 4:])
        print(name, prec, "cls max|diff| %.3e  rel_l2 %.3e" % (np.abs(cls_g - cls_w).max(), rel_l2(got, want)))
        if prec == "fp32":
            assert rel_l2(got, want) <= 2e-4
        else:
            assert rel_l2(got, want) <= 8e-2
        e.close()


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_sppf_pools_fused_launch_equals_three_launches(CE, prec, monkeypatch):
    """SPPF's three chained 5x5 max-pools run as one launch (8-channel slab of a frame's map in LDS, row then column maxima).
    Max-pooling is exact, so every pooled slice and the block output have to equal the three-launch path bit for bit."""
    path, W, g = netutil.model("yolov8n")
    x = netutil.coco_like_frames(3)
    outs = {}
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("ADAS_NO_POOL_FUSE", raising=False)
        else:
            monkeypatch.setenv("ADAS_NO_POOL_FUSE", "1")
        e = CE.HipEngine(path, precision=prec, max_batch=3)
        names = [e.layer_kernel(e.layer_index("model.9.m%d" % i), 3) for i in range(3)]
        assert (("sppf_pool3_kernel" in names[0]) and all("fused into the SPPF" in n for n in names[1:])) == fused, names
        e.engine_inference(x)
        outs[fused] = [e.fetch_activation("model.9.m%d" % i, 3) for i in range(3)] + [e.fetch_activation("model.9.cv2.conv", 3)]
        e.close()
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_neck_upsample_folded_into_its_consumer(CE, prec, monkeypatch):
    """Upsample -> Concat -> C2f.cv1 (YOLO necks): the 1x1 conv reads the upsampled channels from the half-resolution tensor at
    (y / 2, x / 2) and the upsample launch is dropped.  Same values in, same arithmetic: outputs equal the unfolded path bit for bit."""
    path, W, g = netutil.model("yolov8n")
    x = netutil.coco_like_frames(3)
    outs = {}
    for folded in (True, False):
        if folded:
            monkeypatch.delenv("ADAS_NO_UPSAMPLE_FOLD", raising=False)
        else:
            monkeypatch.setenv("ADAS_NO_UPSAMPLE_FOLD", "1")
        e = CE.HipEngine(path, precision=prec, max_batch=3)
        names = [e.layer_kernel(e.layer_index(n), 3) for n in ("model.10", "model.13")]
        assert all(("folded into" in n) == folded for n in names), names
        head = e.engine_inference(x)[0]
        outs[folded] = [e.fetch_activation(n, 3) for n in ("model.12.cv1.conv", "model.15.cv1.conv", "model.15.cv2.conv")] + [head]
        e.close()
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("imgsz", [(640, 640), (352, 608)], ids=str)
@pytest.mark.parametrize("prec,tol", [("fp16", 2e-3), ("bf16", 1.5e-2)])
def test_whole_c2f_block_fused_launch(CE, imgsz, prec, tol):
    """conv_c2f.hip: YOLOv8n's model.2 (cv1 1x1 -> split -> 3x3 Bottleneck pair + shortcut -> cv2 1x1 over the concat) runs as ONE launch
    in the 16-bit modes, the 48-channel concat never written; its output against the oracle's model.2 output, on whole and on ragged
    16 x 16 tiles (88 x 152 map), and the whole head after it."""
    path, W, g = netutil.model("yolov8n", imgsz=imgsz)
    x = netutil.coco_like_frames(2, imgsz[0], imgsz[1], seed=9)
    taps = {}
    want = nets.yolov8_forward(x, W, "n", taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    names = {nm: e.layer_kernel(e.layer_index(nm), 2) for nm in ("model.2.cv1.conv", "model.2.m.0.cv1.conv", "model.2.m.0.cv2.conv", "model.2.cv2.conv")}
    assert names["model.2.cv1.conv"] == "conv_c2f16_kernel" and all("fused into the C2f" in names[k] for k in list(names)[1:]), names
    got = e.engine_inference(x)[0]
    a = e.fetch_activation("model.2.cv2.conv", 2)
    ref = taps["c2f2"].numpy()
    assert a.shape == ref.shape
    print(prec, imgsz, "model.2 out rel_l2 %.3e max|diff| %.3e max|ref| %.2f" % (rel_l2(a, ref), np.abs(a - ref).max(), np.abs(ref).max()))
    assert rel_l2(a, ref) <= tol
    assert rel_l2(got, want) <= 5 * tol
    with pytest.raises(Exception):
        e.fetch_activation("model.2.cv1.conv", 2)       # stays in LDS
    e.close()
