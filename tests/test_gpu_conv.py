"""GPU: single-layer checks of the conv kernels (gather implicit-GEMM and LDS-halo 3x3) through tiny graphs
input(3ch) -> 1x1 expand -> conv under test, against torch conv2d on the same weights.
bf16: rel-L2 <= 1e-2 (two bf16 layers); fp32 mode: max|diff| <= 1e-3."""
import importlib, os, tempfile
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_pkg

pytestmark = pytest.mark.gpu
load_pkg()
M = importlib.import_module("adas_amd.models")


def run_case(CE, H, W, cin, cout, k, s, act, res_mode, prec, batch=2, seed=0, expect_kernel=None, info=None):
    ws = M.SynthWeights(seed, gain=1.0)
    g = M.Graph("unit", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    res = None
    if res_mode != M.RES_NONE:
        ho = (H + 2 * (k // 2) - k) // s + 1
        wo = (W + 2 * (k // 2) - k) // s + 1
        res = g.conv(x, cout, 1, s, "resid", act=M.ACT_NONE, true_cin=c3, pad=0) if (s != 1 or cout != cin) else a
    y = g.conv(a, cout, k, s, "test", act=act, res=res, res_mode=res_mode, f32_out=False)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)       # fp32 tap so engine_inference can return it
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"unit_{H}_{W}_{cin}_{cout}_{k}_{s}_{act}_{res_mode}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    rng = np.random.default_rng(seed)
    xin = rng.uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("test", batch)
    kn = e.layer_kernel(e.layer_index("test"), batch)
    if info is not None:
        info["kernel"] = kn
    if expect_kernel is not None:
        assert expect_kernel in kn, (kn, expect_kernel)
    e.close(); os.remove(path)
    Wt = {k_: torch.from_numpy(v) for k_, v in ws.store.items()}
    with torch.no_grad():
        t = torch.from_numpy(xin)
        a_ = F.silu(F.conv2d(t, Wt["expand.weight"], Wt["expand.bias"]))
        yv = F.conv2d(a_, Wt["test.weight"], Wt["test.bias"], stride=s, padding=k // 2)
        actf = {M.ACT_NONE: lambda v: v, M.ACT_SILU: F.silu, M.ACT_RELU: F.relu, M.ACT_LEAKY: lambda v: F.leaky_relu(v, 0.1)}[act]
        if res_mode != M.RES_NONE:
            r_ = F.conv2d(t, Wt["resid.weight"], Wt["resid.bias"], stride=s) if (s != 1 or cout != cin) else a_
            yv = actf(yv + r_) if res_mode == M.RES_BEFORE_ACT else actf(yv) + r_
        else:
            yv = actf(yv)
    want = yv.numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    rel = float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))
    return rel, float(np.abs(got - want).max())


@pytest.fixture(scope="module")
def CE():
    ce = importlib.import_module("adas_amd.coreEngine")
    assert ce.L.lib().adas_device_count() > 0
    return ce


SHAPES = [(80, 400), (40, 200), (20, 100), (10, 50), (160, 160), (80, 80), (40, 40), (20, 20), (23, 37), (7, 300)]


@pytest.mark.parametrize("hw", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_conv3x3_s1_halo_shapes(CE, hw):
    H, W = hw
    for cin, cout, act, rm in ((64, 64, M.ACT_RELU, M.RES_BEFORE_ACT), (32, 16, M.ACT_SILU, M.RES_NONE),
                               (80, 80, M.ACT_SILU, M.RES_NONE), (128, 32, M.ACT_NONE, M.RES_NONE)):
        rel, mx = run_case(CE, H, W, cin, cout, 3, 1, act, rm, "bf16")
        assert rel < 1e-2, (hw, cin, cout, rel, mx)


@pytest.mark.parametrize("case", [(64, 64, 3, 2), (16, 16, 3, 1), (48, 32, 1, 1), (64, 128, 1, 2), (128, 256, 3, 2),
                                  (256, 64, 3, 1), (24, 40, 3, 1)], ids=str)
def test_conv_gather_variants(CE, case):
    cin, cout, k, s = case
    for prec, tol in (("bf16", 1e-2), ("fp32", 1e-5)):
        rel, mx = run_case(CE, 40, 56, cin, cout, k, s, M.ACT_SILU, M.RES_AFTER_ACT if (s == 1 and cin == cout) else M.RES_NONE, prec)
        assert rel < tol, (case, prec, rel, mx)
        if prec == "fp32":
            assert mx < 1e-3


def test_halo_equals_gather_bitwise_shape(CE, monkeypatch):
    """Same layer through both kernels (ADAS_NO_HALO toggles the dispatcher in a fresh process is not
    possible here, so compare both against the oracle at the tighter 5e-3)."""
    rel, mx = run_case(CE, 80, 80, 64, 64, 3, 1, M.ACT_SILU, M.RES_AFTER_ACT, "bf16")
    assert rel < 5e-3


def run_fc_case(CE, batch, cin, cout, act, f32_out, seed=0, prec="bf16"):
    """input (3,1,1) -> Linear(3, cin)+SiLU -> Linear(cin, cout) under test: both are 1x1-spatial convs and take the
    weight-streaming FC kernel (conv_fc.hip) in bf16 mode."""
    ws = M.SynthWeights(seed, gain=1.0)
    g = M.Graph("fcunit", 3, 1, 1, ws)
    x, c3 = g.input()
    a = g.conv(x, cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    y = g.conv(a, cout, 1, 1, "test", act=act, f32_out=f32_out)
    g.output(y, 0, [1, cout], "o") if f32_out else None
    if not f32_out:
        z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
        g.output(z, 0, [1, 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"fcunit_{batch}_{cin}_{cout}_{act}_{int(f32_out)}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(seed).uniform(-1, 1, (batch, 3, 1, 1)).astype(np.float32)
    out = e.engine_inference(xin)
    got = e.fetch_activation("test", batch).reshape(batch, cout)
    if f32_out:
        assert np.array_equal(out[0].reshape(batch, cout), got)
    e.close(); os.remove(path)
    Wt = {k_: torch.from_numpy(v) for k_, v in ws.store.items()}
    with torch.no_grad():
        t = torch.from_numpy(xin).reshape(batch, 3)
        a_ = F.silu(F.linear(t, Wt["expand.weight"].reshape(cin, 3), Wt["expand.bias"]))
        yv = F.linear(a_, Wt["test.weight"].reshape(cout, cin), Wt["test.bias"])
        yv = {M.ACT_NONE: lambda v: v, M.ACT_SILU: F.silu, M.ACT_RELU: F.relu}[act](yv)
    want = yv.numpy()
    return float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))


@pytest.mark.parametrize("batch", [1, 16, 17, 33, 64])
def test_fc_weight_streaming_kernel(CE, batch):
    # (cin, cout): split-K path (cout <= 8192, K = 4000 = 125 k-steps over 4 waves), wide path with a ragged last tile
    for cin, cout, act, f32 in ((4000, 2048, M.ACT_RELU, False), (2048, 8200, M.ACT_NONE, True), (96, 20, M.ACT_SILU, True)):
        rel = run_fc_case(CE, batch, cin, cout, act, f32)
        assert rel < 1e-2, (batch, cin, cout, rel)


def run_stem_case(CE, H, W, k, pad, cout, act, pool, batch=3, seed=0, prec="bf16"):
    """NCHW fp32 input -> stride-2 kxk conv (+ 3x3 s2 p1 max-pool) through the fused stem kernel (conv_stem.hip)."""
    ws = M.SynthWeights(seed, gain=1.0)
    g = M.Graph("stemunit", 3, H, W, ws)
    x, c3 = g.input()
    y = g.conv(x, cout, k, 2, "stem", act=act, true_cin=c3, pad=pad)
    last = g.maxpool(y, 3, 2, 1, name="pool") if pool else y
    z = g.conv(last, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"stemunit_{H}_{W}_{k}_{cout}_{act}_{int(pool)}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(seed).uniform(-1, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("pool" if pool else "stem", batch)
    if pool:
        with pytest.raises(Exception, match="fused"):
            e.fetch_activation("stem", batch)
    e.close(); os.remove(path)
    Wt = {k_: torch.from_numpy(v) for k_, v in ws.store.items()}
    with torch.no_grad():
        actf = {M.ACT_SILU: F.silu, M.ACT_RELU: F.relu}[act]
        yv = actf(F.conv2d(torch.from_numpy(xin), Wt["stem.weight"], Wt["stem.bias"], stride=2, padding=pad))
        if pool:
            yv = F.max_pool2d(yv, 3, 2, 1)
    want = yv.numpy()
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))


@pytest.mark.parametrize("case", [
    # H, W, k, pad, cout, act, pool
    (64, 128, 7, 3, 64, M.ACT_RELU, True),      # ResNet stem, whole tiles
    (62, 150, 7, 3, 64, M.ACT_RELU, True),      # ragged conv and pool tiles
    (30, 34, 7, 3, 64, M.ACT_RELU, True),       # smaller than one tile
    (64, 64, 3, 1, 16, M.ACT_SILU, False),      # YOLOv8n model.0
    (70, 90, 3, 1, 32, M.ACT_SILU, False),      # YOLOv8s, ragged
    (66, 130, 3, 1, 64, M.ACT_SILU, False),     # YOLOv8l
    (64, 96, 6, 2, 16, M.ACT_SILU, False),      # YOLOv5n model.0 (6x6 s2 p2)
    (62, 70, 6, 2, 32, M.ACT_SILU, False),
    (48, 80, 7, 3, 64, M.ACT_SILU, False),      # 7x7 without a pool behind it
], ids=str)
def test_fused_stem_kernel(CE, case):
    rel = run_stem_case(CE, *case)
    assert rel < 1e-2, (case, rel)


@pytest.mark.parametrize("case", [(32, 32, 1), (48, 32, 1), (80, 80, 1), (96, 64, 1), (192, 64, 1), (384, 128, 1), (512, 256, 1),
                                  (64, 128, 2), (128, 256, 2), (256, 16, 1)], ids=str)
def test_pointwise_kernel(CE, case):
    """1x1 convs (conv_pw.hip): LDS-resident fragment-ordered weights, activations straight into MFMA registers;
    ragged pixel count (23*37*2 is not a multiple of 16), channel tails (48, 80), stride 2."""
    cin, cout, s = case
    for hw in ((23, 37), (40, 56)):
        rel, mx = run_case(CE, hw[0], hw[1], cin, cout, 1, s, M.ACT_SILU, M.RES_NONE, "bf16")
        assert rel < 1e-2, (case, hw, rel, mx)


@pytest.mark.parametrize("case", [(384, 256), (512, 256), (384, 272), (512, 512)], ids=str)
def test_pointwise_kernel_split_output_ranges(CE, case):
    """1x1 convs whose weight matrix exceeds the LDS budget (YOLOv8n model.8/9/21 cv2: 384|512 -> 256): the output tiles are
    split into ranges over blockIdx.y, incl. an uneven split (17 tiles) and a 4-way one."""
    cin, cout = case
    rel, mx = run_case(CE, 20, 20, cin, cout, 1, 1, M.ACT_SILU, M.RES_NONE, "bf16", batch=3, expect_kernel="conv_pw_kernel")
    assert rel < 1e-2, (case, rel, mx)


@pytest.mark.parametrize("hw", [(80, 400), (160, 160), (40, 200), (46, 74), (15, 300), (20, 20)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_conv3x3_s2_halo_shapes(CE, hw):
    """Stride-2 3x3 through the halo kernel (128-pixel tiles): even/odd extents, ragged strips, channel tails."""
    H, W = hw
    for cin, cout, act in ((64, 128, M.ACT_RELU), (32, 64, M.ACT_SILU), (128, 48, M.ACT_NONE), (80, 80, M.ACT_SILU)):
        rel, mx = run_case(CE, H, W, cin, cout, 3, 2, act, M.RES_NONE, "bf16")
        assert rel < 1e-2, (hw, cin, cout, rel, mx)


@pytest.mark.parametrize("case", [(80, 400, 64, 64, M.ACT_RELU, M.RES_BEFORE_ACT), (80, 80, 64, 128, M.ACT_SILU, M.RES_NONE),
                                  (160, 160, 32, 64, M.ACT_SILU, M.RES_NONE), (46, 74, 48, 64, M.ACT_SILU, M.RES_AFTER_ACT),
                                  (80, 80, 16, 128, M.ACT_NONE, M.RES_NONE), (80, 80, 32, 32, M.ACT_SILU, M.RES_AFTER_ACT),
                                  (160, 160, 16, 32, M.ACT_SILU, M.RES_NONE), (46, 74, 64, 24, M.ACT_RELU, M.RES_BEFORE_ACT)], ids=str)
def test_conv3x3_resident_weights_kernel(CE, case):
    """Cin <= 64, Cout > 16 at a batch large enough (>= 1024 tiles) to take the persistent weights-resident kernel
    (conv_halo_rw.hip): several tiles per workgroup, both LDS window buffers, ragged strips, residual modes, and the
    32-output-channel packing (cout <= 32)."""
    H, W, cin, cout, act, rm = case
    rel, mx = run_case(CE, H, W, cin, cout, 3, 1, act, rm, "bf16", batch=max(2, (1024 * 256) // (H * W) + 1),
                       expect_kernel="conv_halo_rw_kernel")
    assert rel < 1e-2, (case, rel, mx)


# ---------------------------------------------------------------------------------------------------------------------
# fp16 (ADAS_PREC_FP16): the same kernels instantiated on the half element tag (elem16.h).  Two half-precision layers in a
# row leave ~5e-4 rel-L2 (11 significant bits); bound 1.5e-3, seven times tighter than the bf16 bound.
FP16_TOL = 1.5e-3


@pytest.mark.parametrize("case", [
    # H, W, cin, cout, k, s, act, res_mode, batch, kernel it must resolve to
    (40, 200, 128, 128, 3, 1, M.ACT_RELU, M.RES_BEFORE_ACT, 2, "conv_halo_kernel"),
    (23, 37, 80, 80, 3, 1, M.ACT_SILU, M.RES_NONE, 2, "conv_halo_kernel"),
    (46, 74, 64, 128, 3, 2, M.ACT_RELU, M.RES_NONE, 2, "conv_halo_kernel"),
    (80, 80, 32, 32, 3, 1, M.ACT_SILU, M.RES_AFTER_ACT, 42, "conv_halo_rw_kernel"),
    (80, 400, 64, 64, 3, 1, M.ACT_RELU, M.RES_BEFORE_ACT, 9, "conv_halo_rw_kernel"),
    (40, 56, 96, 64, 1, 1, M.ACT_SILU, M.RES_NONE, 2, "conv_pw_kernel"),
    (40, 56, 64, 128, 1, 2, M.ACT_SILU, M.RES_NONE, 2, "conv_pw_kernel"),
    (40, 56, 16, 16, 3, 1, M.ACT_SILU, M.RES_AFTER_ACT, 2, None),
], ids=str)
def test_fp16_conv_kernels(CE, case):
    H, W, cin, cout, k, s, act, rm, batch, kern = case
    rel, mx = run_case(CE, H, W, cin, cout, k, s, act, rm, "fp16", batch=batch, expect_kernel=kern)
    print("fp16", case, "rel_l2 %.2e max|diff| %.2e" % (rel, mx))
    assert rel < FP16_TOL, (case, rel, mx)


@pytest.mark.parametrize("batch", [1, 17, 64])
def test_fp16_fc_kernel(CE, batch):
    for cin, cout, act, f32 in ((4000, 2048, M.ACT_RELU, False), (2048, 8200, M.ACT_NONE, True)):
        rel = run_fc_case(CE, batch, cin, cout, act, f32, prec="fp16")
        assert rel < FP16_TOL, (batch, cin, cout, rel)


@pytest.mark.parametrize("case", [(62, 150, 7, 3, 64, M.ACT_RELU, True), (64, 64, 3, 1, 16, M.ACT_SILU, False),
                                  (64, 96, 6, 2, 16, M.ACT_SILU, False), (66, 130, 3, 1, 64, M.ACT_SILU, False)], ids=str)
def test_fp16_fused_stem_kernel(CE, case):
    rel = run_stem_case(CE, *case, prec="fp16")
    assert rel < FP16_TOL, (case, rel)


@pytest.mark.parametrize("case", [
    # H, W (input), cin, cout, act, batch: the ResNet down-sampling convs with Cin >= 128 and YOLO's, ragged extents, a channel
    # tail (160 = 5 chunks), at batches that fill the chip (the kernel is chosen only then)
    (40, 200, 128, 256, M.ACT_RELU, 32), (20, 100, 256, 512, M.ACT_RELU, 72), (40, 40, 128, 256, M.ACT_SILU, 136),
    (23, 37, 160, 256, M.ACT_SILU, 256), (46, 74, 136, 128, M.ACT_NONE, 136),
], ids=str)
@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("fp16", 1.5e-3)])
def test_conv3x3_s2_parity_plane_kernel(CE, case, prec, tol):
    """Stride-2 3x3 with Cout % 128 == 0 and Cin >= 128 through conv_halo_s2.hip (window de-interleaved into four parity planes in
    LDS, 8 waves, 256 output pixels x 128 channels per workgroup)."""
    H, W, cin, cout, act, batch = case
    rel, mx = run_case(CE, H, W, cin, cout, 3, 2, act, M.RES_NONE, prec, batch=batch, expect_kernel="conv_s2p_kernel")
    assert rel < tol, (case, prec, rel, mx)


@pytest.mark.parametrize("case", [
    # H, W, cin, cout, act, res_mode, batch: the ResNet layer2-4 blocks (identity residual), ragged extents with uneven item
    # lists per workgroup and a short last XCD, an odd chunk count (the window buffers swap parity from item to item), a
    # projected residual.  Batches fill the chip in nearly whole rounds of items (the kernel is chosen only then).
    (20, 100, 256, 256, M.ACT_RELU, M.RES_BEFORE_ACT, 16), (10, 50, 512, 512, M.ACT_RELU, M.RES_NONE, 32),
    (40, 200, 128, 128, M.ACT_RELU, M.RES_BEFORE_ACT, 8), (23, 37, 64, 128, M.ACT_SILU, M.RES_NONE, 119),
    (40, 40, 96, 256, M.ACT_SILU, M.RES_AFTER_ACT, 36), (7, 300, 128, 128, M.ACT_NONE, M.RES_NONE, 56),
], ids=str)
@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("fp16", 1.5e-3)])
def test_conv3x3_s1_dma_fed_kernel(CE, case, prec, tol):
    """Stride-1 3x3 with Cout % 128 == 0 through conv_halo8.hip (persistent workgroups, LDS-DMA staging with counted waits, two
    synchronisation variants picked by the chunk count)."""
    H, W, cin, cout, act, res_mode, batch = case
    rel, mx = run_case(CE, H, W, cin, cout, 3, 1, act, res_mode, prec, batch=batch, expect_kernel="conv_h8_kernel")
    assert rel < tol, (case, prec, rel, mx)


@pytest.mark.parametrize("case", [(20, 100, 256, 256, 16), (10, 50, 512, 512, 32), (40, 200, 128, 128, 8)], ids=str)
def test_conv3x3_s1_dma_fed_kernel_is_deterministic(CE, case):
    """Race screen for the counted-wait schedule: a DMA piece read before it landed, or a buffer re-filled while still being read,
    shows up as run-to-run differences.  The same launch, 25 times, has to return identical bits."""
    H, W, cin, cout, batch = case
    ws = M.SynthWeights(3, gain=1.0)
    g = M.Graph("unit", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    y = g.conv(a, cout, 3, 1, "test", act=M.ACT_RELU, res=a, res_mode=M.RES_BEFORE_ACT, f32_out=False)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"unit_det_{H}_{W}_{cin}.hipm")
    g.save(path)
    e = CE.HipEngine(path, "fp16", batch)
    assert "conv_h8_kernel" in e.layer_kernel(e.layer_index("test"), batch)
    xin = np.random.default_rng(5).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    first = e.fetch_activation("test", batch).copy()
    for it in range(24):
        e.engine_inference(xin)
        again = e.fetch_activation("test", batch)
        assert np.array_equal(first, again), (case, it, float(np.abs(first - again).max()))
    e.close(); os.remove(path)


@pytest.mark.parametrize("case", [
    # H, W, cin, cout, stride, act, res_mode: YOLOv8n's class branch (64 / 128 / 256 -> 80 -> 80), a residual, 96 and 72 channels, stride 2
    (80, 80, 64, 80, 1, M.ACT_SILU, M.RES_NONE), (40, 40, 128, 80, 1, M.ACT_SILU, M.RES_NONE), (20, 20, 256, 80, 1, M.ACT_SILU, M.RES_NONE),
    (80, 80, 80, 80, 1, M.ACT_SILU, M.RES_AFTER_ACT), (23, 37, 64, 96, 1, M.ACT_RELU, M.RES_NONE), (40, 40, 96, 72, 1, M.ACT_NONE, M.RES_NONE),
    (46, 74, 64, 80, 2, M.ACT_SILU, M.RES_NONE),
], ids=str)
@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("fp16", 1.5e-3)])
def test_conv3x3_48_wide_blocks(CE, case, prec, tol):
    """65..96 output channels run as two 48-channel blocks (three 16-channel MFMA tiles per wave: the odd tile takes the 8-byte
    store / residual path) instead of two 64-channel blocks with a mostly-padded second block."""
    H, W, cin, cout, s, act, res_mode = case
    rel, mx = run_case(CE, H, W, cin, cout, 3, s, act, res_mode, prec, batch=32, expect_kernel="conv_halo_kernel<48")   # (few tiles would take narrower blocks)
    assert rel < tol, (case, prec, rel, mx)


def _c2f_case(CE, H, W, c2, n, shortcut, prec, batch=3, seed=0):
    """input(3) -> 1x1 expand -> C2f(c2, n) built by models._c2f (channel slices of one concat buffer) -> fp32 tap; torch reference."""
    ws = M.SynthWeights(seed, gain=1.0)
    g = M.Graph("unit", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, c2, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    y = M._c2f(g, a, c2, n, shortcut, "blk")
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"unit_c2f_{H}_{W}_{c2}_{n}_{int(shortcut)}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(seed).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("blk.cv2.conv", batch)
    names = [e.layer_kernel(e.layer_index(f"blk.m.{i}.cv1.conv"), batch) for i in range(n)]
    e.close(); os.remove(path)
    Wt = {k_: torch.from_numpy(v) for k_, v in ws.store.items()}
    cv = lambda t, nm, k: F.silu(F.conv2d(t, Wt[nm + ".weight"], Wt[nm + ".bias"], padding=k // 2))
    with torch.no_grad():
        t = cv(torch.from_numpy(xin), "expand", 1)
        ys = list(cv(t, "blk.cv1.conv", 1).chunk(2, 1))
        for i in range(n):
            b = cv(cv(ys[-1], f"blk.m.{i}.cv1.conv", 3), f"blk.m.{i}.cv2.conv", 3)
            ys.append(ys[-1] + b if shortcut else b)
        want = cv(torch.cat(ys, 1), "blk.cv2.conv", 1).numpy()
    rel = float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))
    return rel, names


@pytest.mark.parametrize("case", [(160, 160, 32, 1, True), (80, 80, 64, 2, True), (80, 80, 64, 1, False), (23, 37, 32, 2, True),
                                  (16, 16, 64, 1, True), (50, 70, 64, 2, False)], ids=str)
@pytest.mark.parametrize("prec,tol", [("bf16", 2e-2), ("fp16", 2.5e-3)])
def test_c2f_bottleneck_pairs_fused(CE, case, prec, tol):
    """The 3x3 -> 3x3 Bottlenecks of a C2f block on 16 / 32 channels run as one launch each (conv_pair.hip): the intermediate stays
    in LDS, the shortcut comes from the staged window; inputs and outputs are channel slices of the block's concat buffer."""
    H, W, c2, n, shortcut = case
    rel, names = _c2f_case(CE, H, W, c2, n, shortcut, prec)
    # a C2f(32, 32, n = 1, shortcut) block is taken as a whole by conv_c2f.hip (round 3): its pair is then part of that launch
    assert all("conv_pair_kernel" in k or "fused into the C2f launch" in k for k in names), names
    assert rel < tol, (case, prec, rel)


@pytest.mark.parametrize("case", [
    # H, W (block output), Cp (block input channels), C, batch: ResNet layer2.0 / layer3.0 / layer4.0 shapes, odd input extents
    (40, 200, 64, 128, 8), (20, 100, 128, 256, 16), (10, 50, 256, 512, 32), (23, 37, 64, 128, 119),
], ids=str)
@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("fp16", 2e-3)])
def test_projection_shortcut_folded_into_conv2(CE, case, prec, tol):
    """ResNet layerN.0: out = relu(conv3x3(t) + conv1x1_s2(x)).  At batches where conv_halo8 takes conv2 the projection runs inside its
    launch (extra K steps on x[2y, 2x] before the 3x3 stream) and the projection conv's own launch is dropped; checked against
    torch, and against the unfolded path's kernel names at a batch too small for the persistent kernel."""
    H, W, cp, c, batch = case
    ws = M.SynthWeights(1, gain=1.0)
    Hin, Win = 2 * H - (H % 2), 2 * W - (W % 2)      # odd block outputs come from odd inputs: (in + 1) // 2
    g = M.Graph("unit", 3, Hin, Win, ws)
    x0, c3 = g.input()
    x = g.conv(x0, cp, 1, 1, "expand", act=M.ACT_RELU, true_cin=c3)
    t = g.conv(x, c, 3, 2, "conv1", act=M.ACT_RELU)
    d = g.conv(x, c, 1, 2, "down", act=M.ACT_NONE, pad=0)
    y = g.conv(t, c, 3, 1, "conv2", act=M.ACT_RELU, res=d, res_mode=M.RES_BEFORE_ACT)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    assert (y.h, y.w) == (H, W)
    path = os.path.join(tempfile.gettempdir(), f"unit_ds_{H}_{W}_{cp}_{c}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(2).uniform(0, 1, (batch, 3, Hin, Win)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("conv2", batch)
    names = (e.layer_kernel(e.layer_index("down"), batch), e.layer_kernel(e.layer_index("conv2"), batch))
    small = (e.layer_kernel(e.layer_index("down"), 1), e.layer_kernel(e.layer_index("conv2"), 1))
    e.close(); os.remove(path)
    assert "shortcut of" in names[0] and "conv_h8_kernel" in names[1] and "+shortcut" in names[1], names
    assert "conv_pw_kernel" in small[0] and "+shortcut" not in small[1], small
    Wt = {k_: torch.from_numpy(v) for k_, v in ws.store.items()}
    with torch.no_grad():
        xt = F.relu(F.conv2d(torch.from_numpy(xin), Wt["expand.weight"], Wt["expand.bias"]))
        tt = F.relu(F.conv2d(xt, Wt["conv1.weight"], Wt["conv1.bias"], stride=2, padding=1))
        dt = F.conv2d(xt, Wt["down.weight"], Wt["down.bias"], stride=2)
        want = F.relu(F.conv2d(tt, Wt["conv2.weight"], Wt["conv2.bias"], padding=1) + dt).numpy()
    rel = float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))
    assert rel < tol, (case, prec, rel)


@pytest.mark.parametrize("k,cin,cout", [(3, 64, 64), (3, 128, 128), (1, 64, 64), (3, 32, 32)], ids=str)
def test_fp16_stores_saturate_instead_of_overflowing(CE, k, cin, cout):
    """Fp16 kernels run with MODE.FP16_OVFL set (elem16.h Fp16::enter): an activation past the half range is stored as +-65504, not
    as inf (which the next layer would turn into NaN).  A conv whose true outputs reach ~1e5..1e6 must come back finite, equal to
    65504 where the oracle exceeds it and equal to the rounded oracle elsewhere."""
    H, W, batch = 40, 40, 2
    rng = np.random.default_rng(3)
    w_exp = np.zeros((cin, 3, 1, 1), np.float32); w_exp[:, :, 0, 0] = rng.uniform(50.0, 300.0, (cin, 3))
    w_test = (rng.uniform(0.0, 2.0, (cout, cin, k, k)) * (9.0 if k == 1 else 1.0)).astype(np.float32)   # same output scale for 1x1 and 3x3
    w_test[: cout // 2] *= 1e-3                                      # half of the channels stay in range
    d = {"expand.weight": w_exp, "expand.bias": np.zeros(cin, np.float32), "test.weight": w_test, "test.bias": np.zeros(cout, np.float32),
         "tap.weight": np.full((8, cout, 1, 1), 1e-6, np.float32), "tap.bias": np.zeros(8, np.float32)}
    g = M.Graph("sat", 3, H, W, M.DictWeights(d))
    x, c3 = g.input()
    a = g.conv(x, cin, 1, 1, "expand", act=M.ACT_NONE, true_cin=c3)
    y = g.conv(a, cout, k, 1, "test", act=M.ACT_NONE)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"sat_{k}_{cin}_{cout}.hipm")
    g.save(path)
    e = CE.HipEngine(path, "fp16", batch)
    xin = rng.uniform(0.2, 1, (batch, 3, H, W)).astype(np.float32)
    out = e.engine_inference(xin)[0]
    got = e.fetch_activation("test", batch)
    e.close(); os.remove(path)
    with torch.no_grad():
        a_ = F.conv2d(torch.from_numpy(xin), torch.from_numpy(w_exp)).half().float()
        want = F.conv2d(a_, torch.from_numpy(w_test).half().float(), padding=k // 2).numpy()
    assert np.isfinite(got).all() and np.isfinite(out).all()
    over = want > 70000.0
    assert over.mean() > 0.2 and (want < 60000.0).mean() > 0.2          # the case really straddles the half range
    assert (got[over] == 65504.0).all()
    inr = want < 60000.0
    assert np.abs(got[inr] - want[inr]).max() <= 2e-3 * np.abs(want[inr]).max()


@pytest.mark.parametrize("case", [(768, 256, 40, 40, 2), (1024, 512, 20, 20, 2), (1280, 512, 20, 20, 16), (2048, 512, 40, 40, 8),
                                  (1152, 576, 23, 37, 3), (768, 80, 20, 20, 2), (400, 320, 40, 40, 2), (224, 160, 80, 80, 2), (328, 64, 20, 20, 2)], ids=str)
@pytest.mark.parametrize("prec,tol", [("fp16", 2e-3), ("bf16", 1e-2)])
def test_wide_pointwise_conv_gemm_kernel(CE, case, prec, tol):
    """conv_pwg.hip: 1x1 convs with Cin > 512 (the C2f / SPPF output convs of YOLOv8 s/m/l/x) and the K-step counts conv_pw is not
    instantiated for (YOLOv8x 160 / 400 channels; a Cin that is no multiple of 32) as a K-looped MFMA GEMM: 128- and 64-pixel tiles,
    ragged pixel, channel and K tails, with SiLU."""
    cin, cout, H, W, batch = case
    rel, mx = run_case(CE, H, W, cin, cout, 1, 1, M.ACT_SILU, M.RES_NONE, prec, batch=batch, expect_kernel="conv_pwg_kernel")
    assert rel < tol, (case, prec, rel, mx)


def test_wide_pointwise_conv_with_residual(CE):
    rel, mx = run_case(CE, 20, 20, 1024, 1024, 1, 1, M.ACT_SILU, M.RES_AFTER_ACT, "fp16", batch=2, expect_kernel="conv_pwg_kernel")
    assert rel < 2e-3, (rel, mx)


@pytest.mark.parametrize("name,batch", [("yolov8s", 1), ("yolov8s", 64), ("yolov8m", 16), ("yolov8l", 1), ("yolov8l", 16), ("yolov8x", 4), ("yolov10n", 64), ("yolov10s", 16), ("yolov9t", 64), ("yolov9s", 16), ("yolov9c", 8),
                                        ("yolov7-tiny", 64), ("yolov6n", 64), ("yolov6s", 16)])
def test_no_generic_fallback_kernel_in_16bit_modes(CE, name, batch):
    """Every conv of the YOLOv8 s / m / l / x graphs (and YOLOv10n's plain convs) resolves to a specialised kernel in the 16-bit
    modes: conv_igemm_kernel (the generic implicit GEMM) is the fp32 parity path and the shapes nothing else takes.  The only
    exception allowed: 1x1 convs WITH a residual (PSA's proj / ffn.1 in YOLOv10n, Cin 128 / 256)."""
    import netutil
    path, W, g = netutil.model(name)
    e = CE.HipEngine(path, "fp16", batch)
    ops = {o["name"]: o for o in g.ops}
    bad = []
    for i in range(e.stats()["num_layers"]):
        k = e.layer_kernel(i, batch)
        nm = e.layer_info(i)[0]
        if "conv_igemm" in k and not (ops[nm]["kh"] == 1 and ops[nm]["res"] is not None):
            bad.append((nm, k))
    e.close()
    assert not bad, bad
