"""GPU: the split precision (ADAS_PREC_FP16X3, "fp16x3"): every value a (hi, lo) pair of halves, every product three f16 MFMAs with
fp32 accumulation (csrc/elem16.h, conv_x3.hip).  Its contract is the fp32 parity mode's: f32-class layer results, every discrete
decision of the fp32 oracle chain reproduced (BASELINE.json north_star: "bit-exact for NMS survivor indices / ByteTrack ID
assignment", yoloDetector.py:126-157, byteTracker.py:62-185) -- on the 16-bit matrix cores instead of the 1/16-rate f32 MFMA.

Layer tests bound the error against torch fp32 at 20x below anything a half-precision layer can reach (fp16: ~6e-4 rel-L2) and print
it next to the fp32 mode's error on the same case; network tests use the fp32 mode's bounds of tests/test_gpu_configs.py.
"""
import importlib
import os

import numpy as np
import pytest

import netutil
from conftest import load_pkg
from oracle import nets

import test_gpu_conv as TC
import test_gpu_configs as TG

pytestmark = pytest.mark.gpu
load_pkg()
M = importlib.import_module("adas_amd.models")
CE = importlib.import_module("adas_amd.coreEngine")

X3_REL = 3e-6      # rel-L2 of one tested layer behind a 1x1 expand layer (measured values are printed)


@pytest.mark.parametrize("case", [(64, 64, 3, 1, M.ACT_RELU, M.RES_BEFORE_ACT), (64, 128, 3, 2, M.ACT_RELU, M.RES_NONE),
                                  (16, 16, 3, 1, M.ACT_SILU, M.RES_AFTER_ACT), (48, 32, 1, 1, M.ACT_SILU, M.RES_NONE),
                                  (64, 128, 1, 2, M.ACT_NONE, M.RES_NONE), (256, 64, 3, 1, M.ACT_SILU, M.RES_NONE),
                                  (24, 40, 3, 1, M.ACT_LEAKY, M.RES_NONE), (128, 128, 3, 1, M.ACT_RELU, M.RES_BEFORE_ACT),
                                  (8, 64, 7, 2, M.ACT_RELU, M.RES_NONE)], ids=str)
def test_conv_layers_f32_class(case):
    cin, cout, k, s, act, rm = case
    info = {}
    rel, mx = TC.run_case(CE, 40, 56, cin, cout, k, s, act, rm, "fp16x3", info=info)
    rel32, mx32 = TC.run_case(CE, 40, 56, cin, cout, k, s, act, rm, "fp32")
    print("x3 %s: rel %.2e max %.2e  (fp32 mode: rel %.2e max %.2e)  %s" % (case, rel, mx, rel32, mx32, info.get("kernel")))
    assert "x3" in info["kernel"], info
    assert rel < X3_REL and mx < 1e-4, (case, rel, mx)


@pytest.mark.parametrize("case", [(32, 32, 1, M.ACT_SILU), (40, 80, 1, M.ACT_SILU), (96, 64, 1, M.ACT_SILU), (192, 64, 1, M.ACT_SILU),
                                  (128, 256, 2, M.ACT_NONE), (160, 160, 1, M.ACT_LEAKY), (384, 128, 1, M.ACT_SILU), (512, 256, 1, M.ACT_SILU),
                                  (256, 512, 2, M.ACT_NONE)], ids=str)
def test_pointwise_layers_on_the_streaming_kernel(case):
    """conv_pw_x3.hip: weights resident in LDS as hi | lo fragment blocks, activations straight from the G8 tensor; K steps 1..16, the
    ragged last step (40 channels), stride 2 (the ResNet projections) and the layers whose weights are cut into ranges."""
    cin, cout, s, act = case
    info = {}
    rel, mx = TC.run_case(CE, 40, 56, cin, cout, 1, s, act, M.RES_NONE, "fp16x3", info=info, batch=3)
    print("x3 pointwise %s: rel %.2e max %.2e  %s" % (case, rel, mx, info.get("kernel")))
    assert "conv_pwx3_kernel" in info["kernel"], info
    assert rel < X3_REL and mx < 1e-4, (case, rel, mx)


@pytest.mark.parametrize("hw", [(80, 400), (23, 37), (7, 300), (20, 20)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_conv3x3_shapes(hw):
    H, W = hw
    for cin, cout, act, rm in ((64, 64, M.ACT_RELU, M.RES_BEFORE_ACT), (32, 16, M.ACT_SILU, M.RES_NONE), (80, 80, M.ACT_SILU, M.RES_NONE)):
        rel, mx = TC.run_case(CE, H, W, cin, cout, 3, 1, act, rm, "fp16x3")
        assert rel < X3_REL, (hw, cin, cout, rel, mx)


@pytest.mark.parametrize("case", [(1, 4000, 2048, M.ACT_RELU, False), (64, 2048, 1000, M.ACT_NONE, True), (3, 512, 91224, M.ACT_NONE, True),
                                  (130, 256, 264, M.ACT_NONE, True), (40, 512, 9000, M.ACT_NONE, True), (20, 256, 8200, M.ACT_RELU, False)], ids=str)
def test_linear_layers_f32_class(case):
    batch, cin, cout, act, f32_out = case
    rel = TC.run_fc_case(CE, batch, cin, cout, act, f32_out, prec="fp16x3")
    rel = rel[0] if isinstance(rel, tuple) else rel
    print("x3 linear %s: rel %.2e" % (case, rel))
    assert rel < X3_REL, (case, rel)


def test_tiny_and_huge_magnitudes_survive_the_split():
    """lo is scaled by 2^11 and hi is zeroed below the half normal range (the value moves into lo, keeping 11 bits): values from the
    half normal range up to 3e4 keep 22 bits, and whatever lies below it is off by at most 2^-14 * 2^-11 relative to NOTHING larger
    than itself -- an absolute floor of ~3e-8 * 2^-11.  A conv on inputs ~0.1 with weights ~0.3 x He, on O(1) and on O(300) inputs comes
    out with f32-class relative error; on inputs ~1e-4 with weights ~1e-3 (every value below the half normal range) the error is
    bounded absolutely."""
    import os, tempfile
    import torch
    import torch.nn.functional as F
    for in_scale, w_scale in ((0.1, 0.3), (1.0, 1.0), (300.0, 0.5), (1e-4, 1e-3)):
        ws = M.SynthWeights(5, gain=1.0)
        g = M.Graph("mag", 3, 24, 40, ws)
        x, c3 = g.input()
        a = g.conv(x, 32, 1, 1, "expand", act=M.ACT_NONE, true_cin=c3)
        y = g.conv(a, 32, 3, 1, "test", act=M.ACT_NONE)
        g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
        for nm in list(ws.store):                      # weights and biases of both layers
            if nm.startswith(("expand.", "test.")):
                ws.store[nm] = ws.store[nm] * np.float32(w_scale if nm.endswith("weight") else w_scale * in_scale)
        # (the first graph only drew the seeded arrays: the container is built from the scaled ones)
        g2 = M.Graph("mag", 3, 24, 40, M.DictWeights(dict(ws.store)))
        x, c3 = g2.input()
        a = g2.conv(x, 32, 1, 1, "expand", act=M.ACT_NONE, true_cin=c3)
        y = g2.conv(a, 32, 3, 1, "test", act=M.ACT_NONE)
        z = g2.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
        g2.output(z, 0, [1, z.h * z.w * 8], "o")
        path = os.path.join(tempfile.gettempdir(), "x3_mag.hipm")
        g2.save(path)
        e = CE.HipEngine(path, "fp16x3", 2)
        xin = (np.random.default_rng(1).uniform(-1, 1, (2, 3, 24, 40)) * in_scale).astype(np.float32)
        e.engine_inference(xin)
        got = e.fetch_activation("test", 2)
        e.close(); os.remove(path)
        Wt = {k_: torch.from_numpy(v).double() for k_, v in ws.store.items() if k_.startswith(("expand.", "test."))}
        with torch.no_grad():
            t = torch.from_numpy(xin).double()
            want = F.conv2d(F.conv2d(t, Wt["expand.weight"], Wt["expand.bias"]), Wt["test.weight"], Wt["test.bias"], padding=1).numpy()
        rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        mx = float(np.abs(got - want).max())
        print("x3 magnitudes in %.0e w %.0e: rel %.2e  max|diff| %.2e  max|ref| %.2e" % (in_scale, w_scale, rel, mx, np.abs(want).max()))
        if in_scale * w_scale > 1e-3:
            assert rel < X3_REL, (in_scale, w_scale, rel)
        else:
            assert mx < 1e-9, (in_scale, w_scale, mx)


@pytest.mark.parametrize("backbone", ["18", "34"])
def test_ufldv2_culane_full_geometry_vs_oracle(backbone):
    """C3 at full geometry (stem 7x7 s2, 3x3 s2 max-pool, residual stages, 1x1 pool conv, LayerNorm, both Linear layers): the fp32
    mode's bound (max|diff| <= 1e-3) and rel-L2 <= 1e-5."""
    path, W, g = netutil.model("ufldv2_res" + backbone)
    x = netutil.lane_frames(2, 320, 1600, seed=7)
    taps = {}
    want = nets.ufldv2_forward(x, W, backbone, taps=taps)
    e = CE.HipEngine(path, precision="fp16x3", max_batch=2)
    got = e.engine_inference(x)
    last = "model.layer4.%d.conv2" % (1 if backbone == "18" else 2)
    a = e.fetch_activation(last, 2)
    ref = taps["layer4"].numpy()
    err4, rel4 = TG.report("ufldv2-r%s fp16x3 layer4" % backbone, a, ref)
    assert err4 <= 1e-3 * max(1.0, float(np.abs(ref).max())) and rel4 <= 1e-5
    for o, w in zip(got, want):
        err, rel = TG.report("ufldv2-r%s fp16x3 output" % backbone, o, w)
        assert err <= 1e-3 * max(1.0, float(np.abs(w).max())) and rel <= 1e-5
    e.close()


@pytest.mark.parametrize("scale", ["n", "s"])
def test_yolov8_640_vs_oracle(tmp_path, scale):
    """C2 / C4 detectors with a calibrated class branch: the fp32 mode's bounds."""
    x = netutil.coco_like_frames(2, seed=11)
    path, W = TG.calibrated(tmp_path, "yolov8" + scale, x, "x3_%s" % scale)
    taps = {}
    want = nets.yolov8_forward(x, W, scale, taps=taps)
    e = CE.HipEngine(path, precision="fp16x3", max_batch=2)
    got = e.engine_inference(x)[0]
    for lname, key in (("model.15.cv2.conv", "p3"), ("model.18.cv2.conv", "p4"), ("model.21.cv2.conv", "p5")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err, rel = TG.report("yolov8%s fp16x3 %s" % (scale, key), a, ref)
        assert err <= 1e-3 * max(1.0, float(np.abs(ref).max())) and rel <= 1e-5, lname
    errh, relh = TG.report("yolov8%s fp16x3 head" % scale, got, want)
    ecls = float(np.abs(got[:, 4:] - want[:, 4:]).max())
    ebox = float(np.abs(got[:, :4] - want[:, :4]).max())
    print("yolov8%s fp16x3 max|prob diff| %.3e  max|box diff| %.3e px" % (scale, ecls, ebox))
    assert relh <= 1e-4 and ecls <= 1e-4 and ebox <= 1e-3 * max(1.0, float(np.abs(want[:, :4]).max()))
    kernels = [e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])]
    got = np.array(got, copy=True)
    e.close()
    # the head above came from ONE launch (detect_v8_fused_x3_kernel: the six closing 1x1 convs + the decode); the separate launches agree
    assert any("detect_v8_fused_x3_kernel" in k for k in kernels) and sum("fused into the Detect launch" in k for k in kernels) == 6, kernels
    os.environ["ADAS_NO_DETECT_FUSE"] = "1"
    try:
        e = CE.HipEngine(path, precision="fp16x3", max_batch=2)
    finally:
        del os.environ["ADAS_NO_DETECT_FUSE"]
    sep = np.array(e.engine_inference(x)[0], copy=True)
    assert not any("detect_v8_fused" in e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"]))
    e.close()
    dc, db = float(np.abs(got[:, 4:] - sep[:, 4:]).max()), float(np.abs(got[:, :4] - sep[:, :4]).max())
    print("yolov8%s fp16x3 fused Detect vs separate launches: max|prob diff| %.2e  max|box diff| %.2e px" % (scale, dc, db))
    assert dc <= 2e-6 and db <= 2e-3      # the same (hi, lo) products, summed in a different order


def test_yolov10n_vs_oracle():
    """Depth-wise convs and PSA attention in the split storage (fp32 arithmetic on the joined values): the head against the oracle, with
    the fp32 mode's bounds (tests/test_gpu_v10.py) and next to the fp32 mode's own error on the same frames."""
    path, W, g = netutil.model("yolov10n")
    x = netutil.coco_like_frames(2, seed=4)
    want = nets.detector_forward("yolov10n", x, W)
    res = {}
    for prec in ("fp32", "fp16x3"):
        e = CE.HipEngine(path, precision=prec, max_batch=2)
        got = np.array(e.engine_inference(x)[0], copy=True)
        e.close()
        err, rel = TG.report("yolov10n %s head" % prec, got, want)
        res[prec] = (rel, float(np.abs(got[:, 4:] - want[:, 4:]).max()), float(np.abs(got[:, :4] - want[:, :4]).max()), got)
        print("yolov10n %s: rel %.2e  max|prob diff| %.2e  max|box diff| %.2e px" % ((prec,) + res[prec][:3]))
    print("yolov10n fp16x3 vs fp32 mode: rel %.2e" % TG.rel_l2(res["fp16x3"][3], res["fp32"][3]))
    rel, ecls, ebox, _ = res["fp16x3"]
    assert ecls <= 1e-3 and ebox <= 1e-3 * max(1.0, float(np.abs(want[:, :4]).max()))
    assert rel <= max(1e-4, 3 * res["fp32"][0])


@pytest.mark.parametrize("case", [(80, 400, 64, 64, M.ACT_RELU, M.RES_BEFORE_ACT, 2), (40, 200, 128, 128, M.ACT_RELU, M.RES_BEFORE_ACT, 8),
                                  (80, 400, 32, 64, M.ACT_SILU, M.RES_NONE, 2), (40, 56, 256, 64, M.ACT_NONE, M.RES_NONE, 100),
                                  (40, 56, 64, 128, M.ACT_LEAKY, M.RES_AFTER_ACT, 64), (23, 37, 64, 64, M.ACT_SILU, M.RES_AFTER_ACT, 128),
                                  (80, 80, 80, 80, M.ACT_SILU, M.RES_NONE, 64), (80, 80, 64, 80, M.ACT_SILU, M.RES_NONE, 64),
                                  (80, 80, 32, 32, M.ACT_SILU, M.RES_AFTER_ACT, 64), (40, 40, 128, 80, M.ACT_SILU, M.RES_NONE, 64),
                                  (40, 40, 48, 192, M.ACT_RELU, M.RES_NONE, 64)], ids=str)
def test_halo8_x3_kernel_layers(case):
    """The persistent LDS-DMA kernel of the split precision (conv_halo8_x3.hip: half-chunk stream, main / cross accumulators) at batches
    that fill the chip: whole and ragged maps (strip tiles wrap rows, window rows outside the image zero-filled by the DMA), with
    and without a residual, one to three 64-channel blocks per tile, channel counts that are not whole blocks (80 -> 80: Cin padded to
    96 by zero weight columns, Cout to 128 by zero rows with masked stores)."""
    H, W, cin, cout, act, rm, batch = case
    info = {}
    rel, mx = TC.run_case(CE, H, W, cin, cout, 3, 1, act, rm, "fp16x3", batch=batch, info=info)
    print("h8x3 %s: rel %.2e max %.2e  %s" % (case, rel, mx, info.get("kernel")))
    assert "conv_h8x3_kernel" in info["kernel"], info
    assert rel < X3_REL and mx < 1e-4, (case, rel, mx)


@pytest.mark.parametrize("case", [(10, 50, 512, 512, 3, 1, M.ACT_RELU, M.RES_BEFORE_ACT, 1), (20, 100, 256, 256, 3, 1, M.ACT_RELU, M.RES_NONE, 1),
                                  (20, 100, 256, 512, 3, 2, M.ACT_RELU, M.RES_NONE, 1), (20, 20, 256, 80, 3, 1, M.ACT_SILU, M.RES_NONE, 1),
                                  (20, 20, 128, 128, 3, 1, M.ACT_SILU, M.RES_AFTER_ACT, 2), (13, 17, 96, 40, 3, 1, M.ACT_LEAKY, M.RES_NONE, 3),
                                  (7, 9, 64, 200, 5, 1, M.ACT_NONE, M.RES_NONE, 1)], ids=str)
def test_small_maps_on_the_k_split_kernel(case):
    """conv_x3_ksplit_kernel (one frame at a time: the K loop split over the eight waves of a workgroup, fragments straight from global
    memory, partial tiles summed in a fixed tree): UFLDv2's layer3 / layer4 shapes at one frame, YOLOv8n's 20x20 Detect convs, a Cout that
    is not a multiple of the 32-column tile (80, 40, 200), ragged pixel tiles, a 5x5 kernel, every residual mode -- f32-class against
    torch fp32, and the same bits on a second run (the reduction order is fixed)."""
    H, W, cin, cout, k, stride, act, rm, batch = case
    info = {}
    rel, mx = TC.run_case(CE, H, W, cin, cout, k, stride, act, rm, "fp16x3", batch=batch, info=info)
    print("x3 k-split %s: rel %.2e max %.2e  %s" % (case, rel, mx, info.get("kernel")))
    assert "conv_x3_ksplit_kernel" in info["kernel"], info
    assert rel < X3_REL and mx < 1e-4, (case, rel, mx)
    rel2, mx2 = TC.run_case(CE, H, W, cin, cout, k, stride, act, rm, "fp16x3", batch=batch)
    assert (rel2, mx2) == (rel, mx)


@pytest.mark.parametrize("case", [(80, 400, 64, 128, M.ACT_RELU, 8), (40, 200, 128, 256, M.ACT_RELU, 16), (20, 100, 256, 512, M.ACT_RELU, 64),
                                  (160, 160, 32, 64, M.ACT_SILU, 16), (80, 80, 64, 128, M.ACT_SILU, 64), (40, 40, 128, 256, M.ACT_SILU, 64),
                                  (80, 80, 64, 64, M.ACT_LEAKY, 64), (46, 74, 64, 80, M.ACT_NONE, 64), (23, 37, 96, 192, M.ACT_SILU, 128)], ids=str)
@pytest.mark.parametrize("form", ["dma", "dma-persistent", "registers"])
def test_s2p_x3_kernel_layers(case, form, monkeypatch):
    """The stride-2 3x3 conv of the split precision on the parity-plane kernels of conv_halo_s2.hip (half-chunk window, conv_halo8_x3's weight
    slabs, main / cross accumulators): the three UFLD down-sampling layers, the YOLOv8n ones, even and odd map sizes (the last window row /
    column outside the image), a Cout that is not whole 64-channel blocks (80: zero weight rows, masked stores), three chunks (96 input
    channels) -- f32-class against torch fp32.  Three forms: conv_s2d_x3_kernel (window and weights by LDS-DMA, one parity plane recycled at
    a time; the default) launched fine-grained and as one persistent workgroup per CU (ADAS_S2D_ROUNDS=1: every workgroup walks several
    items, the next item's first half-chunk arrives under the last taps), and the register-staged conv_s2p_x3_kernel (ADAS_NO_S2D_X3=1;
    ADAS_NO_HALO_S2P_X3=1 sends these layers back to the generic kernel).  The switches are read once per process: the non-default
    forms run in a child process."""
    H, W, cin, cout, act, batch = case
    if form != "dma":
        import subprocess, sys, json
        env = dict(os.environ, **({"ADAS_S2D_ROUNDS": "1"} if form == "dma-persistent" else {"ADAS_NO_S2D_X3": "1"}))
        code = ("import sys, json, importlib; sys.path.insert(0, %r); sys.path.insert(0, %r); from conftest import load_pkg; load_pkg();"
                "import test_gpu_conv as TC; CE = importlib.import_module('adas_amd.coreEngine'); info = {};"
                "rel, mx = TC.run_case(CE, %d, %d, %d, %d, 3, 2, %d, 0, 'fp16x3', batch=%d, info=info); print(json.dumps([rel, mx, info.get('kernel')]))"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), H, W, cin, cout, act, batch))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        rel, mx, kernel = json.loads(r.stdout.strip().splitlines()[-1])
    else:
        info = {}
        rel, mx = TC.run_case(CE, H, W, cin, cout, 3, 2, act, M.RES_NONE, "fp16x3", batch=batch, info=info)
        kernel = info.get("kernel")
    print("s2 x3 [%s] %s: rel %.2e max %.2e  %s" % (form, case, rel, mx, kernel))
    assert ("conv_s2p_x3_kernel" if form == "registers" else "conv_s2d_x3_kernel") in kernel, kernel
    assert rel < X3_REL and mx < 1e-4, (case, rel, mx)


def test_ufldv2_culane_at_the_bench_batch_vs_oracle():
    """The C3 network at batch 64 (the bench's batch): the 3x3 stride-1 layers run on conv_h8x3_kernel.  Three distinct frames tiled
    over the batch: frames 0-2 against the oracle, every copy bit-identical to the first (different workgroups, same arithmetic)."""
    path, W, g = netutil.model("ufldv2_res18")
    x3 = netutil.lane_frames(3, 320, 1600, seed=11)
    x = np.ascontiguousarray(np.concatenate([x3] * 22, 0)[:64])
    want = nets.ufldv2_forward(x3, W, "18")
    e = CE.HipEngine(path, precision="fp16x3", max_batch=64)
    kernels = [e.layer_kernel(i, 64) for i in range(e.stats()["num_layers"])]
    assert sum("conv_h8x3_kernel" in k for k in kernels) >= 13, kernels
    got = e.engine_inference(x)
    for o, w in zip(got, want):
        err, rel = TG.report("ufldv2-r18 batch 64 fp16x3 output", o[:3], w)
        assert rel <= 1e-5 and err <= 1e-3 * max(1.0, float(np.abs(w).max()))
        for k in range(3, 64):
            assert np.array_equal(o[k], o[k % 3]), (k, float(np.abs(o[k] - o[k % 3]).max()))
    e.close()


@pytest.mark.parametrize("case", [(160, 160, 32, 1, True, 4), (23, 37, 32, 1, True, 3), (48, 50, 32, 1, True, 70), (80, 80, 64, 2, True, 2)], ids=str)
def test_c2f_block_in_one_launch_f32_class(case):
    """conv_c2f_x3.hip: a C2f(32, 32, n = 1, shortcut) block (YOLOv8n / YOLOv10n model.2) runs as ONE launch in the split precision -- hi / lo
    planes in LDS between the stages, three MFMAs per product, exact SiLU -- and comes out f32-class against torch; ragged extents (tiles
    cut by the image edge), more tiles than the persistent launch has workgroups (48 x 50 at batch 70: 1,120 tiles on 256 workgroups) and
    a block the fusion does NOT cover (64 channels, two Bottlenecks: its pairs are released back to the per-layer kernels)."""
    H, W, c2, n, shortcut, batch = case
    rel, names = TC._c2f_case(CE, H, W, c2, n, shortcut, "fp16x3", batch=batch)
    print("x3 C2f %s: rel %.2e  %s" % (case, rel, names))
    if (c2, n) == (32, 1):
        assert all("fused into the C2f launch" in k for k in names), names
    else:
        assert all("fused" not in k and "x3" in k for k in names), names
    assert rel < 3 * X3_REL, (case, rel)


@pytest.mark.parametrize("case", [(640, 640, 3, 1, 2), (70, 90, 3, 1, 3), (96, 200, 6, 2, 40), (33, 47, 6, 2, 2)], ids=str)
def test_stem_with_its_second_conv_in_one_launch_f32_class(case):
    """conv_stem2_x3_kernel: the YOLO stem (3x3 / 6x6 s2, 3 -> 16, SiLU) and the 3x3 s2 conv behind it (16 -> 32, SiLU: model.1) as one
    launch in the split precision -- the stem tile lives in LDS as a hi and a lo plane, the 16-channel tensor never reaches HBM.  Full
    and ragged extents (odd stem / conv2 sizes: tiles cut by both images' edges), more tiles than workgroups (96 x 200 at batch 40)."""
    import os, tempfile
    import torch
    import torch.nn.functional as F
    H, W, k, pad, batch = case
    ws = M.SynthWeights(3, gain=1.0)
    g = M.Graph("stem2unit", 3, H, W, ws)
    x, c3 = g.input()
    y = g.conv(x, 16, k, 2, "stem", act=M.ACT_SILU, true_cin=c3, pad=pad)
    t = g.conv(y, 32, 3, 2, "second", act=M.ACT_SILU)
    z = g.conv(t, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"stem2unit_{H}_{W}_{k}.hipm")
    g.save(path)
    e = CE.HipEngine(path, "fp16x3", batch)
    xin = np.random.default_rng(4).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("second", batch)
    names = (e.layer_kernel(e.layer_index("stem"), batch), e.layer_kernel(e.layer_index("second"), batch))
    e.close(); os.remove(path)
    assert "conv_stem2_x3_kernel" in names[0] and "fused into the stem launch" in names[1], names
    Wt = {k_: torch.from_numpy(v) for k_, v in ws.store.items()}
    with torch.no_grad():
        s1 = F.silu(F.conv2d(torch.from_numpy(xin), Wt["stem.weight"], Wt["stem.bias"], stride=2, padding=pad))
        want = F.silu(F.conv2d(s1, Wt["second.weight"], Wt["second.bias"], stride=2, padding=1)).numpy()
    rel = float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))
    print("x3 stem + second conv %s: rel %.2e" % (case, rel))
    assert got.shape == want.shape and rel < 3 * X3_REL, (case, rel)
