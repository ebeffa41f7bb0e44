"""Seeded parameters of a torchvision-topology ResNet-18/34 + UFLD head, in the reference modules' OWN state_dict naming
(conv / BatchNorm tensors un-folded), and the fold into the '<conv>.weight' / '<conv>.bias' dict oracle/nets.py and models.py
take.  Shared by tests/golden/make_golden_ufldnet.py (which loads them into the reference's parsingNet) and
tests/test_oracle_golden.py (which regenerates them from the seed: the 780 MB full-geometry head is never stored).
No reference dependency."""
import math
import numpy as np

RESNET_DEPTHS = {"18": [2, 2, 2, 2], "34": [3, 4, 6, 3]}
BN_EPS = 1e-5   # torch.nn.BatchNorm2d default, what torchvision's ResNet uses


def _bn(rng, c):
    return dict(weight=rng.uniform(0.6, 1.4, c).astype(np.float32), bias=(0.1 * rng.standard_normal(c)).astype(np.float32),
                running_mean=(0.1 * rng.standard_normal(c)).astype(np.float32), running_var=rng.uniform(0.5, 1.5, c).astype(np.float32))


def resnet_state(rng, depth, gain=0.8):
    """'model.*' entries of the reference's `resnet` wrapper (backbone.py:40-47 re-exports conv1/bn1/layer1..4)."""
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * gain * math.sqrt(2.0 / (cin * k * k))).astype(np.float32)

    def bn(name, c):
        for k, v in _bn(rng, c).items():
            sd[name + "." + k] = v

    conv("model.conv1", 64, 3, 7)
    bn("model.bn1", 64)
    cin = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], RESNET_DEPTHS[depth])):
        for bi in range(nblk):
            s = 2 if (li > 0 and bi == 0) else 1
            base = "model.layer%d.%d" % (li + 1, bi)
            conv(base + ".conv1", planes, cin, 3)
            bn(base + ".bn1", planes)
            conv(base + ".conv2", planes, planes, 3)
            bn(base + ".bn2", planes)
            if s != 1 or cin != planes:
                conv(base + ".downsample.0", planes, cin, 1)
                bn(base + ".downsample.1", planes)
            cin = planes
    return sd


def _linear(rng, sd, name, cout, cin):
    sd[name + ".weight"] = (rng.standard_normal((cout, cin), dtype=np.float32) * np.float32(math.sqrt(2.0 / cin)))
    sd[name + ".bias"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)


def ufldv2_state(seed, depth, in_h, in_w, grid_row, cls_row, grid_col, cls_col, lanes=4, fc_norm=True):
    """state_dict (numpy) of exportLib/ultrafastLaneV2/model_culane.parsingNet (model_culane.py:7-40)."""
    rng = np.random.default_rng(seed)
    sd = resnet_state(rng, depth)
    sd["pool.weight"] = (rng.standard_normal((8, 512, 1, 1)) * math.sqrt(1.0 / 512)).astype(np.float32)
    sd["pool.bias"] = (0.05 * rng.standard_normal(8)).astype(np.float32)
    input_dim = in_h // 32 * in_w // 32 * 8
    total = (grid_row * cls_row + grid_col * cls_col + 2 * cls_row + 2 * cls_col) * lanes
    if fc_norm:
        sd["cls.0.weight"] = (1.0 + 0.05 * rng.standard_normal(input_dim)).astype(np.float32)
        sd["cls.0.bias"] = (0.05 * rng.standard_normal(input_dim)).astype(np.float32)
    _linear(rng, sd, "cls.1", 2048, input_dim)
    _linear(rng, sd, "cls.3", total, 2048)
    return sd


def ufld1_state(seed, depth, griding_num, cls_per_lane, lanes=4):
    """state_dict (numpy) of exportLib/ultrafastLane/model.parsingNet (model.py:19-69; 288x800 input, 1800-wide flatten)."""
    rng = np.random.default_rng(seed)
    sd = resnet_state(rng, depth)
    sd["pool.weight"] = (rng.standard_normal((8, 512, 1, 1)) * math.sqrt(1.0 / 512)).astype(np.float32)
    sd["pool.bias"] = (0.05 * rng.standard_normal(8)).astype(np.float32)
    _linear(rng, sd, "cls.0", 2048, 1800)
    _linear(rng, sd, "cls.2", (griding_num + 1) * cls_per_lane * lanes, 2048)
    return sd


def fold(sd):
    """BatchNorm folded into the preceding conv: the name -> array dict models.build / oracle.nets consume
    (conv 'X.convN' + bn 'X.bnN'; 'downsample.0' + 'downsample.1'; 'model.conv1' + 'model.bn1')."""
    out = {}
    for k, w in sd.items():
        if not k.endswith(".weight") or w.ndim != 4:
            continue
        base = k[:-len(".weight")]
        if base.endswith(".downsample.0"):
            bn = base[:-1] + "1"
        elif base[-5:-1] == "conv":
            bn = base[:-5] + "bn" + base[-1]
        else:
            bn = None
        b = sd.get(base + ".bias")
        if bn is not None and bn + ".running_mean" in sd:
            g, beta, mean, var = (sd[bn + s].astype(np.float64) for s in (".weight", ".bias", ".running_mean", ".running_var"))
            scale = g / np.sqrt(var + BN_EPS)
            out[base + ".weight"] = (w.astype(np.float64) * scale.reshape(-1, 1, 1, 1)).astype(np.float32)
            b0 = np.zeros(w.shape[0]) if b is None else b.astype(np.float64)
            out[base + ".bias"] = ((b0 - mean) * scale + beta).astype(np.float32)
        else:
            out[base + ".weight"] = w
            out[base + ".bias"] = np.zeros(w.shape[0], np.float32) if b is None else b
    for k, v in sd.items():
        if k.startswith("cls."):
            out[k] = v
    return out


def lane_frame(seed, h, w):
    """One normalised NCHW fp32 frame (ImageNet mean/std, ultrafastLaneDetectorV2.py:104-110) with low-frequency structure."""
    rng = np.random.default_rng(seed)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    x = rng.integers(0, 255, (1, 3, h, w)).astype(np.float32) / 255.0
    x = 0.5 * x + 0.5 * np.repeat(np.repeat(rng.uniform(0, 1, (1, 3, h // 16, w // 16)).astype(np.float32), 16, 2), 16, 3)
    return ((x - mean) / std).astype(np.float32)


# (tag, kind, depth, geometry kwargs): the reduced case, the two BASELINE geometries (configs/culane_res18.py, culane_res34.py),
# the Tusimple head (no LayerNorm, configs/tusimple_res18.py) and UFLD v1 (ultrafastLaneDetector.py:16-40 Tusimple config)
CASES = [
    ("v2_r18_small", "v2", "18", dict(in_h=160, in_w=800, grid_row=100, cls_row=36, grid_col=50, cls_col=41, fc_norm=True)),
    ("v2_r18_culane", "v2", "18", dict(in_h=320, in_w=1600, grid_row=200, cls_row=72, grid_col=100, cls_col=81, fc_norm=True)),
    ("v2_r34_culane", "v2", "34", dict(in_h=320, in_w=1600, grid_row=200, cls_row=72, grid_col=100, cls_col=81, fc_norm=True)),
    ("v2_r18_tusimple", "v2", "18", dict(in_h=320, in_w=800, grid_row=100, cls_row=56, grid_col=100, cls_col=41, fc_norm=False)),
    ("v1_r18_tusimple", "v1", "18", dict(griding_num=100, cls_per_lane=56)),
    # CurveLanes configuration (configs/curvelanes_res18.py: 72/41 anchors, 10 lanes, LayerNorm) at a reduced input
    ("v2_r18_curvelanes_small", "v2", "18", dict(in_h=256, in_w=512, grid_row=200, cls_row=72, grid_col=100, cls_col=41, lanes=10, fc_norm=True)),
]
SEED = 20240
SAMPLE = 4096   # values kept per output tensor (evenly strided over the flattened tensor)


def sample_idx(n):
    return np.unique(np.linspace(0, n - 1, min(n, SAMPLE)).astype(np.int64))
