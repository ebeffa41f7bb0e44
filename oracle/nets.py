"""Oracle: torch-CPU fp32 forward passes of the detector / lane networks.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinning status:
  * UFLDv2 / UFLD v1 forward: PINNED to the reference's own network modules (exportLib/ultrafastLaneV2/model_culane.py,
    exportLib/ultrafastLane/model.py, imported unmodified under a torchvision stub and fed seeded un-folded parameters;
    tests/golden/make_golden_ufldnet.py -> tests/golden/ufld_net.npz; tests/test_oracle_golden.py compares at <= 2e-5 of the
    output range, measured ~1e-6): reduced geometry, CULane R18 and R34 at 1600x320, the Tusimple head, UFLD v1.
  * YOLOv8 / YOLOv5 forward: PARITY UNPINNED -- the reference has no weights and no YOLO architecture; onnxruntime (its CPU
    path, coreEngine.py:159-186) is not installed.
These functions restate the architectures the reference's exported models come from and stand in for "ONNXRuntime-CPU"
(oneDNN fp32):
  * YOLOv8: ultralytics 8.1.x (README.md:56) Conv/C2f/SPPF/Detect/DFL; output layout pinned by
    ObjectDetector/yoloDetector.py:110-122 -> (1, 4+nc, 8400) [cx,cy,w,h,probs] in input pixels.
  * YOLOv5 v6.2 (README.md:53) Conv/C3/SPPF/Detect; output (1, 25200, 5+nc), anchors
    yoloDetector.py:23.
  * UFLDv2: TrafficLaneDetector/ufldDetector/exportLib/ultrafastLaneV2/model_culane.py:7-63 on the
    torchvision ResNet topology used by backbone.py:14-58 (BasicBlock, BN folded into conv+bias).
Weights are a plain dict name -> ndarray (BatchNorm pre-folded: '<conv>.weight' OIHW, '<conv>.bias').
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


def _t(W, name):
    return torch.from_numpy(np.ascontiguousarray(W[name]))


# None | "fp16" | "bf16": storage-rounding emulation of the 16-bit engine modes on the CPU (weights and every conv output rounded
# to the storage type, fp32 accumulate; head logits (act=None) stay fp32 like the engine's f32 head buffers).  A study / expectation
# tool for tests and tools/ (how much of a 16-bit deviation is storage rounding): the parity reference is always EMULATE = None.
EMULATE = None


def _round(t):
    if EMULATE == "fp16":
        return t.half().float()
    if EMULATE == "bf16":
        return t.bfloat16().float()
    return t


def _conv(x, W, name, s=1, p=None, act="silu"):
    w = _round(_t(W, name + ".weight"))
    b = _t(W, name + ".bias")
    k = w.shape[-1]
    y = F.conv2d(x, w, b, stride=s, padding=(k // 2 if p is None else p))
    if act == "silu":
        y = _round(F.silu(y))
    elif act == "relu":
        y = _round(F.relu(y))
    elif act == "leaky":
        y = _round(F.leaky_relu(y, 0.1))
    return y


# ------------------------------------------------------------------ YOLOv8
V8_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.0, 1.0, 512),
             "x": (1.0, 1.25, 512)}


def _c2f(x, W, name, n, shortcut):
    y = list(_conv(x, W, f"{name}.cv1.conv").chunk(2, 1))
    for i in range(n):
        t = _conv(_conv(y[-1], W, f"{name}.m.{i}.cv1.conv"), W, f"{name}.m.{i}.cv2.conv")
        y.append(y[-1] + t if shortcut else t)
    return _conv(torch.cat(y, 1), W, f"{name}.cv2.conv")


def _sppf(x, W, name):
    x = _conv(x, W, f"{name}.cv1.conv")
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    return _conv(torch.cat((x, y1, y2, y3), 1), W, f"{name}.cv2.conv")


def yolov8_forward(x, W, scale="n", nc=80, taps=None):
    """x: (N,3,H,W) fp32 in [0,1].  Returns (N, 4+nc, A) fp32.  taps: optional dict filled with named activations."""
    depth = V8_SCALES[scale][0]
    dep = lambda n: max(round(n * depth), 1)
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        x = _conv(x, W, "model.0.conv", 2)
        x = _conv(x, W, "model.1.conv", 2)
        x2 = _c2f(x, W, "model.2", dep(3), True)
        x = _conv(x2, W, "model.3.conv", 2)
        x4 = _c2f(x, W, "model.4", dep(6), True)
        x = _conv(x4, W, "model.5.conv", 2)
        x6 = _c2f(x, W, "model.6", dep(6), True)
        x = _conv(x6, W, "model.7.conv", 2)
        x = _c2f(x, W, "model.8", dep(3), True)
        x9 = _sppf(x, W, "model.9")
        if taps is not None:
            taps.update(c2f2=x2)
        x = torch.cat((F.interpolate(x9, scale_factor=2, mode="nearest"), x6), 1)
        x12 = _c2f(x, W, "model.12", dep(3), False)
        x = torch.cat((F.interpolate(x12, scale_factor=2, mode="nearest"), x4), 1)
        x15 = _c2f(x, W, "model.15", dep(3), False)
        x = torch.cat((_conv(x15, W, "model.16.conv", 2), x12), 1)
        x18 = _c2f(x, W, "model.18", dep(3), False)
        x = torch.cat((_conv(x18, W, "model.19.conv", 2), x9), 1)
        x21 = _c2f(x, W, "model.21", dep(3), False)
        if taps is not None:
            taps.update(p3=x15, p4=x18, p5=x21, b4=x4, b6=x6, sppf=x9)
        feats = [x15, x18, x21]
        N = x.shape[0]
        outs = []
        for i, f in enumerate(feats):
            b = _conv(_conv(f, W, f"model.22.cv2.{i}.0.conv"), W, f"model.22.cv2.{i}.1.conv")
            b = _conv(b, W, f"model.22.cv2.{i}.2", act=None)
            c = _conv(_conv(f, W, f"model.22.cv3.{i}.0.conv"), W, f"model.22.cv3.{i}.1.conv")
            c = _conv(c, W, f"model.22.cv3.{i}.2", act=None)
            outs.append(torch.cat((b, c), 1).view(N, 64 + nc, -1))
        return _v8_decode(outs, [f.shape[2:] for f in feats], nc, taps)


def _v8_decode(outs, sizes, nc, taps=None):
    """ultralytics Detect inference path: per-level (N, 64+nc, h*w) logits -> (N, 4+nc, A): DFL expectation over 16 bins, dist2bbox
    (xywh) x stride, sigmoid class probabilities."""
    N = outs[0].shape[0]
    anchors = []
    for h, w in sizes:
        sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
        anchors.append(torch.stack((sx, sy), -1).view(-1, 2))
    H_in = sizes[0][0] * 8
    stride_t = torch.cat([torch.full((h * w, 1), float(H_in // h)) for h, w in sizes])
    xcat = torch.cat(outs, 2)
    box, cls = xcat.split((64, nc), 1)
    A = box.shape[2]
    d = box.view(N, 4, 16, A).transpose(2, 1).softmax(1)
    dist = (d * torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)).sum(1)      # (N,4,A)
    anc = torch.cat(anchors).transpose(0, 1).unsqueeze(0)                                # (1,2,A)
    lt, rb = dist.chunk(2, 1)
    x1y1 = anc - lt
    x2y2 = anc + rb
    dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * stride_t.transpose(0, 1)
    if taps is not None:
        taps.update(box_logits=box, cls_logits=cls)
    return torch.cat((dbox, cls.sigmoid()), 1).numpy()


# ------------------------------------------------------------------ YOLOv10
# THU-MIG/yolov10 (ultralytics 8.1 fork) yolov10n.yaml + nn/modules/block.py (SCDown, PSA/Attention, CIB, C2fCIB, RepVGGDW fused to one
# 7x7) + head.py v10Detect (one-to-one branch).  PARITY UNPINNED like the other YOLO graphs (architecture not in the reference; the
# reference only pins the I/O: demo.py:24-30 ships yolov10n, yoloDetector.py:114,121 decodes a v8-layout (1, 4+nc, A) tensor);
# parameter / FLOP counts match the published 2.3 M / 6.7 G.
V10_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024)}      # yolov10s.yaml differs from n in row 8 (C2fCIB with the 7x7 branch)


def _dwconv(x, W, name, s=1, act="silu"):
    w = _round(_t(W, name + ".weight"))
    k = w.shape[-1]
    y = F.conv2d(x, w, _t(W, name + ".bias"), stride=s, padding=k // 2, groups=x.shape[1])
    return _round(F.silu(y)) if act == "silu" else _round(y)


def _scdown(x, W, name, s=2):
    return _dwconv(_conv(x, W, f"{name}.cv1.conv"), W, f"{name}.cv2.conv", s, act=None)


def _psa(x, W, name):
    c = x.shape[1] // 2
    nh = c // 64
    hd = c // nh
    kd = hd // 2
    a, b = _conv(x, W, f"{name}.cv1.conv").split((c, c), 1)
    B, _, H, Wd = b.shape
    N = H * Wd
    qkv = _round(_conv(b, W, f"{name}.attn.qkv.conv", act=None))
    q, k, v = qkv.view(B, nh, 2 * kd + hd, N).split([kd, kd, hd], dim=2)
    attn = ((q.transpose(-2, -1) @ k) * (kd ** -0.5)).softmax(dim=-1)
    y = _round((v @ attn.transpose(-2, -1)).reshape(B, c, H, Wd))
    y = _round(y + _dwconv(v.reshape(B, c, H, Wd), W, f"{name}.attn.pe.conv", act=None))
    b = _round(b + _conv(y, W, f"{name}.attn.proj.conv", act=None))
    b = _round(b + _conv(_conv(b, W, f"{name}.ffn.0.conv"), W, f"{name}.ffn.1.conv", act=None))
    return _conv(torch.cat((a, b), 1), W, f"{name}.cv2.conv")


def _cib(x, W, name, lk):
    t = _dwconv(x, W, f"{name}.cv1.0.conv")
    t = _conv(t, W, f"{name}.cv1.1.conv")
    t = _dwconv(t, W, f"{name}.cv1.2.conv.conv" if lk else f"{name}.cv1.2.conv")      # deploy form of RepVGGDW: one 7x7 + SiLU
    t = _conv(t, W, f"{name}.cv1.3.conv")
    return x + _dwconv(t, W, f"{name}.cv1.4.conv")


def _c2fcib(x, W, name, n, lk):
    y = list(_conv(x, W, f"{name}.cv1.conv").chunk(2, 1))
    for i in range(n):
        y.append(_cib(y[-1], W, f"{name}.m.{i}", lk))
    return _conv(torch.cat(y, 1), W, f"{name}.cv2.conv")


def yolov10_forward(x, W, scale="n", nc=80, taps=None):
    """x: (N,3,H,W) fp32 in [0,1] -> (N, 4+nc, A) fp32 in the v8 head layout the reference decodes (yoloDetector.py:114,121)."""
    depth = V10_SCALES[scale][0]
    dep = lambda n: max(round(n * depth), 1)
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        x = _conv(x, W, "model.0.conv", 2)
        x = _conv(x, W, "model.1.conv", 2)
        x = _c2f(x, W, "model.2", dep(3), True)
        x = _conv(x, W, "model.3.conv", 2)
        x4 = _c2f(x, W, "model.4", dep(6), True)
        x = _scdown(x4, W, "model.5")
        x6 = _c2f(x, W, "model.6", dep(6), True)
        x = _scdown(x6, W, "model.7")
        x = _c2f(x, W, "model.8", dep(3), True) if scale == "n" else _c2fcib(x, W, "model.8", dep(3), True)
        x = _sppf(x, W, "model.9")
        x10 = _psa(x, W, "model.10")
        x = torch.cat((F.interpolate(x10, scale_factor=2, mode="nearest"), x6), 1)
        x13 = _c2f(x, W, "model.13", dep(3), False)
        x = torch.cat((F.interpolate(x13, scale_factor=2, mode="nearest"), x4), 1)
        x16 = _c2f(x, W, "model.16", dep(3), False)
        x = torch.cat((_conv(x16, W, "model.17.conv", 2), x13), 1)
        x19 = _c2f(x, W, "model.19", dep(3), False)
        x = torch.cat((_scdown(x19, W, "model.20"), x10), 1)
        x22 = _c2fcib(x, W, "model.22", dep(3), True)
        if taps is not None:
            taps.update(p3=x16, p4=x19, p5=x22, psa=x10)
        outs = []
        feats = [x16, x19, x22]
        N = x.shape[0]
        for i, f in enumerate(feats):
            b = _conv(_conv(f, W, f"model.23.one2one_cv2.{i}.0.conv"), W, f"model.23.one2one_cv2.{i}.1.conv")
            b = _conv(b, W, f"model.23.one2one_cv2.{i}.2", act=None)
            c = _conv(_dwconv(f, W, f"model.23.one2one_cv3.{i}.0.0.conv"), W, f"model.23.one2one_cv3.{i}.0.1.conv")
            c = _conv(_dwconv(c, W, f"model.23.one2one_cv3.{i}.1.0.conv"), W, f"model.23.one2one_cv3.{i}.1.1.conv")
            c = _conv(c, W, f"model.23.one2one_cv3.{i}.2", act=None)
            outs.append(torch.cat((b, c), 1).view(N, 64 + nc, -1))
        return _v8_decode(outs, [f.shape[2:] for f in feats], nc, taps)


# ------------------------------------------------------------------ YOLOv9t
# GELAN-t (ultralytics yolov9t.yaml: "917 layers, 2128720 parameters, 8.5 GFLOPs" un-fused; this deploy form with RepConv
# re-parameterised to one 3x3 counts 2.09 M / 8.2 G): ELAN1, AConv, RepNCSPELAN4 (RepCSP of RepBottlenecks), SPPELAN, v8 Detect.
# PARITY UNPINNED like the other YOLO graphs (README.md:57 lists YOLOv9; yoloDetector.py:114,121 pins the head layout only).
def _repcsp(x, W, name, n):
    y = _conv(x, W, f"{name}.cv1.conv")
    for i in range(n):
        y = _round(y + _conv(_conv(y, W, f"{name}.m.{i}.cv1.conv"), W, f"{name}.m.{i}.cv2.conv"))
    return _conv(torch.cat((y, _conv(x, W, f"{name}.cv2.conv")), 1), W, f"{name}.cv3.conv")


def _repncspelan4(x, W, name, n):
    y = list(_conv(x, W, f"{name}.cv1.conv").chunk(2, 1))
    y.append(_conv(_repcsp(y[-1], W, f"{name}.cv2.0", n), W, f"{name}.cv2.1.conv"))
    y.append(_conv(_repcsp(y[-1], W, f"{name}.cv3.0", n), W, f"{name}.cv3.1.conv"))
    return _conv(torch.cat(y, 1), W, f"{name}.cv4.conv")


def _elan1(x, W, name):
    y = list(_conv(x, W, f"{name}.cv1.conv").chunk(2, 1))
    y.append(_conv(y[-1], W, f"{name}.cv2.conv"))
    y.append(_conv(y[-1], W, f"{name}.cv3.conv"))
    return _conv(torch.cat(y, 1), W, f"{name}.cv4.conv")


def _aconv(x, W, name):
    return _conv(_round(F.avg_pool2d(x, 2, 1, 0, False, True)), W, f"{name}.cv1.conv", 2)


def _sppelan(x, W, name):
    y = [_conv(x, W, f"{name}.cv1.conv")]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    return _conv(torch.cat(y, 1), W, f"{name}.cv5.conv")


def yolov9t_forward(x, W, nc=80, taps=None):
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        x = _conv(x, W, "model.0.conv", 2)
        x = _conv(x, W, "model.1.conv", 2)
        x = _elan1(x, W, "model.2")
        x = _aconv(x, W, "model.3")
        x4 = _repncspelan4(x, W, "model.4", 3)
        x = _aconv(x4, W, "model.5")
        x6 = _repncspelan4(x, W, "model.6", 3)
        x = _aconv(x6, W, "model.7")
        x = _repncspelan4(x, W, "model.8", 3)
        x9 = _sppelan(x, W, "model.9")
        x = torch.cat((F.interpolate(x9, scale_factor=2, mode="nearest"), x6), 1)
        x12 = _repncspelan4(x, W, "model.12", 3)
        x = torch.cat((F.interpolate(x12, scale_factor=2, mode="nearest"), x4), 1)
        x15 = _repncspelan4(x, W, "model.15", 3)
        x = torch.cat((_aconv(x15, W, "model.16"), x12), 1)
        x18 = _repncspelan4(x, W, "model.18", 3)
        x = torch.cat((_aconv(x18, W, "model.19"), x9), 1)
        x21 = _repncspelan4(x, W, "model.21", 3)
        if taps is not None:
            taps.update(p3=x15, p4=x18, p5=x21, sppelan=x9)
        feats = [x15, x18, x21]
        N = x.shape[0]
        outs = []
        for i, f in enumerate(feats):
            b = _conv(_conv(f, W, f"model.22.cv2.{i}.0.conv"), W, f"model.22.cv2.{i}.1.conv")
            b = _conv(b, W, f"model.22.cv2.{i}.2", act=None)
            c = _conv(_conv(f, W, f"model.22.cv3.{i}.0.conv"), W, f"model.22.cv3.{i}.1.conv")
            c = _conv(c, W, f"model.22.cv3.{i}.2", act=None)
            outs.append(torch.cat((b, c), 1).view(N, 64 + nc, -1))
        return _v8_decode(outs, [f.shape[2:] for f in feats], nc, taps)


def _adown(x, W, name):
    """ultralytics nn/modules/block.py ADown.forward."""
    x = _round(F.avg_pool2d(x, 2, 1, 0, False, True))
    x1, x2 = x.chunk(2, 1)
    return torch.cat((_conv(x1, W, f"{name}.cv1.conv", 2), _conv(F.max_pool2d(x2, 3, 2, 1), W, f"{name}.cv2.conv")), 1)


def yolov9c_forward(x, W, nc=80, taps=None):
    """ultralytics cfg/models/v9/yolov9c.yaml row by row (RepConv in its fused deploy form)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        x = _conv(x, W, "model.0.conv", 2)
        x = _conv(x, W, "model.1.conv", 2)
        x = _repncspelan4(x, W, "model.2", 1)
        x = _adown(x, W, "model.3")
        x4 = _repncspelan4(x, W, "model.4", 1)
        x = _adown(x4, W, "model.5")
        x6 = _repncspelan4(x, W, "model.6", 1)
        x = _adown(x6, W, "model.7")
        x = _repncspelan4(x, W, "model.8", 1)
        x9 = _sppelan(x, W, "model.9")
        x = torch.cat((F.interpolate(x9, scale_factor=2, mode="nearest"), x6), 1)
        x12 = _repncspelan4(x, W, "model.12", 1)
        x = torch.cat((F.interpolate(x12, scale_factor=2, mode="nearest"), x4), 1)
        x15 = _repncspelan4(x, W, "model.15", 1)
        x = torch.cat((_adown(x15, W, "model.16"), x12), 1)
        x18 = _repncspelan4(x, W, "model.18", 1)
        x = torch.cat((_adown(x18, W, "model.19"), x9), 1)
        x21 = _repncspelan4(x, W, "model.21", 1)
        if taps is not None:
            taps.update(p3=x15, p4=x18, p5=x21, sppelan=x9)
        feats = [x15, x18, x21]
        N = x.shape[0]
        outs = []
        for i, f in enumerate(feats):
            b = _conv(_conv(f, W, f"model.22.cv2.{i}.0.conv"), W, f"model.22.cv2.{i}.1.conv")
            b = _conv(b, W, f"model.22.cv2.{i}.2", act=None)
            c = _conv(_conv(f, W, f"model.22.cv3.{i}.0.conv"), W, f"model.22.cv3.{i}.1.conv")
            c = _conv(c, W, f"model.22.cv3.{i}.2", act=None)
            outs.append(torch.cat((b, c), 1).view(N, 64 + nc, -1))
        return _v8_decode(outs, [f.shape[2:] for f in feats], nc, taps)


# ------------------------------------------------------------------ YOLOv6 v3.0 n / s
# meituan/YOLOv6: yolov6/models/efficientrep.py (EfficientRep), reppan.py (RepBiFPANNeck), effidehead.py (Detect.forward, eval),
# layers/common.py (RepVGGBlock deploy = 3x3 conv + ReLU, ConvBNReLU, SimCSPSPPF, BiFusion, Transpose), configs/yolov6n.py / yolov6s.py
# (depth 0.33, width 0.25 / 0.50, num_repeats [1, 6, 12, 18, 6] + [12, 12, 12, 12], fuse_P2, cspsppf, use_dfl False).
V6_SCALES = {"n": (0.33, 0.25), "s": (0.33, 0.50)}


def _v6_rep(x, W, name, s=1):
    return _conv(x, W, name + ".rbr_reparam", s, act="relu")


def _v6_cbr(x, W, name, s=1):
    return _conv(x, W, name + ".block.conv", s, act="relu")


def _v6_repblock(x, W, name, n):
    x = _v6_rep(x, W, name + ".conv1")
    for i in range(n - 1):
        x = _v6_rep(x, W, f"{name}.block.{i}")
    return x


def _v6_simcspsppf(x, W, name):
    x1 = _v6_cbr(_v6_cbr(_v6_cbr(x, W, name + ".cv1"), W, name + ".cv3"), W, name + ".cv4")
    y0 = _v6_cbr(x, W, name + ".cv2")
    y1 = F.max_pool2d(x1, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = _v6_cbr(_v6_cbr(torch.cat((x1, y1, y2, F.max_pool2d(y2, 5, 1, 2)), 1), W, name + ".cv5"), W, name + ".cv6")
    return _v6_cbr(torch.cat((y0, y3), 1), W, name + ".cv7")


def _v6_bifusion(xs, W, name):
    wt = _round(_t(W, name + ".upsample.upsample_transpose.weight"))
    x0 = _round(F.conv_transpose2d(xs[0], wt, _t(W, name + ".upsample.upsample_transpose.bias"), stride=2))
    x1 = _v6_cbr(xs[1], W, name + ".cv1")
    x2 = _v6_cbr(_v6_cbr(xs[2], W, name + ".cv2"), W, name + ".downsample", 2)
    return _v6_cbr(torch.cat((x0, x1, x2), 1), W, name + ".cv3")


def yolov6_forward(x, W, scale="n", nc=80, taps=None):
    """x: (N,3,H,W) fp32 in [0,1] -> (N, A, 5+nc): [cx, cy, w, h] in input pixels, objectness 1, class probabilities."""
    depth = V6_SCALES[scale][0]
    rn = lambda n: max(round(n * depth), 1) if n > 1 else n
    nb, nk = [rn(n) for n in (1, 6, 12, 18, 6)], rn(12)
    x = torch.as_tensor(x, dtype=torch.float32)
    H_in = x.shape[2]
    with torch.no_grad():
        x = _v6_rep(x, W, "backbone.stem", 2)
        outs = []
        for i in range(1, 5):
            x = _v6_repblock(_v6_rep(x, W, f"backbone.ERBlock_{i + 1}.0", 2), W, f"backbone.ERBlock_{i + 1}.1", nb[i])
            outs.append(x)
        x3, x2, x1, _ = outs
        x0 = _v6_simcspsppf(x, W, "backbone.ERBlock_5.2")
        fpn_out0 = _v6_cbr(x0, W, "neck.reduce_layer0")
        f_out0 = _v6_repblock(_v6_bifusion([fpn_out0, x1, x2], W, "neck.Bifusion0"), W, "neck.Rep_p4", nk)
        fpn_out1 = _v6_cbr(f_out0, W, "neck.reduce_layer1")
        pan_out2 = _v6_repblock(_v6_bifusion([fpn_out1, x2, x3], W, "neck.Bifusion1"), W, "neck.Rep_p3", nk)
        pan_out1 = _v6_repblock(torch.cat((_v6_cbr(pan_out2, W, "neck.downsample2", 2), fpn_out1), 1), W, "neck.Rep_n3", nk)
        pan_out0 = _v6_repblock(torch.cat((_v6_cbr(pan_out1, W, "neck.downsample1", 2), fpn_out0), 1), W, "neck.Rep_n4", nk)
        if taps is not None:
            taps.update(p3=pan_out2, p4=pan_out1, p5=pan_out0, sppf=x0)
        cls_l, reg_l, pts, strd = [], [], [], []
        for i, f in enumerate((pan_out2, pan_out1, pan_out0)):
            b, _, h, w = f.shape
            st = _conv(f, W, f"detect.stems.{i}.conv")
            cls_l.append(torch.sigmoid(_conv(_conv(st, W, f"detect.cls_convs.{i}.conv"), W, f"detect.cls_preds.{i}", act=None)).reshape(b, nc, h * w))
            reg_l.append(_conv(_conv(st, W, f"detect.reg_convs.{i}.conv"), W, f"detect.reg_preds.{i}", act=None).reshape(b, 4, h * w))
            sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
            pts.append(torch.stack((sx, sy), -1).reshape(-1, 2))
            strd.append(torch.full((h * w, 1), float(H_in // h)))
        cls_score = torch.cat(cls_l, -1).permute(0, 2, 1)
        dist = torch.cat(reg_l, -1).permute(0, 2, 1)
        anchor_points, stride_tensor = torch.cat(pts), torch.cat(strd)
        x1y1, x2y2 = anchor_points - dist[..., :2], anchor_points + dist[..., 2:]
        boxes = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), -1) * stride_tensor
        return torch.cat((boxes, torch.ones(boxes.shape[0], boxes.shape[1], 1), cls_score), -1).numpy()


# ------------------------------------------------------------------ EfficientDet-D0
# PARITY UNPINNED (the reference ships no EfficientDet weights or graph; it loads an exported efficientdet-d0 .onnx through
# onnxruntime, ObjectDetector/efficientdetDetector.py:18-44).  Restates the published architecture: EfficientNet-B0 (arXiv:1905.11946
# table 1: MBConv with squeeze-and-excitation 0.25, swish), BiFPN with fast normalised fusion (arXiv:1911.09070 section 3.3, eq. 3,
# D0: 64 channels x 3 cells) and the shared separable-conv class / box heads (section 4: 3 layers at D0, 9 anchors), BatchNorm folded.
EFFNET_B0 = [(1, 3, 1, 16, 1), (6, 3, 2, 24, 2), (6, 5, 2, 40, 2), (6, 3, 2, 80, 3), (6, 5, 1, 112, 3), (6, 5, 2, 192, 4), (6, 3, 1, 320, 1)]


def _fusion(W, name):
    w = np.maximum(np.asarray(W[name], np.float32), np.float32(0))
    return (w / (w.sum(dtype=np.float32) + np.float32(1e-4))).astype(np.float32)


def _se(x, W, name):
    m = x.mean((2, 3), keepdim=True)
    h = F.silu(F.conv2d(m, _t(W, name + ".reduce.weight"), _t(W, name + ".reduce.bias")))
    return _round(x * torch.sigmoid(F.conv2d(h, _t(W, name + ".expand.weight"), _t(W, name + ".expand.bias"))))


def _mbconv(x, W, name, e, k, s, cout):
    cin = x.shape[1]
    t = _conv(x, W, name + ".expand") if e != 1 else x
    t = _dwconv(t, W, name + ".dw", s)
    t = _se(t, W, name + ".se")
    t = _conv(t, W, name + ".project", act=None)
    return _round(t + x) if (s == 1 and cin == cout) else _round(t)


def _sepconv(x, W, name, act=None, dw_name=None):
    w = _round(_t(W, (dw_name or name) + ".dw.weight"))
    t = _round(F.conv2d(x, w, None, padding=1, groups=x.shape[1]))
    y = _conv(t, W, name + ".pw", act=act)
    return y if act is None and name.endswith(".header") else _round(y)


def efficientdet_forward(x, W, nc=90, fpn_cells=3, head_layers=3, taps=None):
    """-> (regression (N, A, 4) rows (level, y, x, anchor) as (dy, dx, dh, dw), class logits (N, A, nc))."""
    tap = (lambda k, v: taps.__setitem__(k, v.numpy().copy())) if taps is not None else (lambda k, v: None)
    t = _conv(x, W, "stem", s=2)
    feats, bi = [], 0
    for si, (e, k, s, c, n) in enumerate(EFFNET_B0):
        for r in range(n):
            t = _mbconv(t, W, f"blocks.{bi}", e, k, s if r == 0 else 1, c)
            bi += 1
        if si in (2, 4, 6):
            feats.append(t)
    c3, c4, c5 = feats
    tap("c3", c3); tap("c4", c4); tap("c5", c5)
    up = lambda v: F.interpolate(v, scale_factor=2, mode="nearest")
    down = lambda v: F.max_pool2d(v, 3, 2, 1)
    p = None
    for cell in range(fpn_cells):
        nm = f"bifpn.{cell}"
        if cell == 0:
            p3 = _conv(c3, W, nm + ".p3_down", act=None); p4 = _conv(c4, W, nm + ".p4_down", act=None); p5 = _conv(c5, W, nm + ".p5_down", act=None)
            p3, p4, p5 = _round(p3), _round(p4), _round(p5)
            p6 = down(_round(_conv(c5, W, nm + ".p5_to_p6", act=None)))
            p7 = down(p6)
            p4b, p5b = _round(_conv(c4, W, nm + ".p4_down_2", act=None)), _round(_conv(c5, W, nm + ".p5_down_2", act=None))
        else:
            p3, p4, p5, p6, p7 = p
            p4b, p5b = p4, p5

        def fuse(tag, ins):
            w = _fusion(W, f"{nm}.{tag}")
            acc = 0
            for wi, v in zip(w, ins):
                acc = acc + float(wi) * v
            return _round(F.silu(acc))

        p6u = _sepconv(fuse("p6_w1", [p6, up(p7)]), W, nm + ".conv6_up")
        p5u = _sepconv(fuse("p5_w1", [p5, up(p6u)]), W, nm + ".conv5_up")
        p4u = _sepconv(fuse("p4_w1", [p4, up(p5u)]), W, nm + ".conv4_up")
        p3o = _sepconv(fuse("p3_w1", [p3, up(p4u)]), W, nm + ".conv3_up")
        p4o = _sepconv(fuse("p4_w2", [p4b, p4u, down(p3o)]), W, nm + ".conv4_down")
        p5o = _sepconv(fuse("p5_w2", [p5b, p5u, down(p4o)]), W, nm + ".conv5_down")
        p6o = _sepconv(fuse("p6_w2", [p6, p6u, down(p5o)]), W, nm + ".conv6_down")
        p7o = _sepconv(fuse("p7_w2", [p7, down(p6o)]), W, nm + ".conv7_down")
        p = (p3o, p4o, p5o, p6o, p7o)
        for lv, v in enumerate(p):
            tap(f"bifpn{cell}.p{lv + 3}", v)
    regs, clss = [], []
    for lv, f in enumerate(p):
        for branch, dst, per in (("regressor", regs, 4), ("classifier", clss, nc)):
            t = f
            for i in range(head_layers):
                t = _sepconv(t, W, f"{branch}.l{lv}.{i}", act="silu", dw_name=f"{branch}.conv_list.{i}")
            o = _sepconv(t, W, f"{branch}.l{lv}.header", dw_name=f"{branch}.header")
            dst.append(o.permute(0, 2, 3, 1).reshape(o.shape[0], -1, per))
    return torch.cat(regs, 1), torch.cat(clss, 1)


def head_layout(name):
    """"yolov5" (A, 5+nc) or "yolov8" (4+nc, A): the `model_type` argument of oracle.yolo_post.detect_post for graph `name`."""
    return "yolov5" if name.startswith(("yolov5", "yolov6", "yolov7")) else "yolov8"


def detector_forward(name, x, W, nc=80, taps=None):
    """Forward of the detector graph `name`: (4+nc, A) heads ("yolov8n" .. "yolov8x", "yolov10n", "yolov10s", "yolov9t", "yolov9s", "yolov9c") or the v5 layout
    (A, 5+nc) ("yolov7-tiny", "yolov6n", "yolov6s", "yolov5n" .. "yolov5x"; no taps for YOLOv5)."""
    if name.startswith("yolov7"):
        return yolov7_tiny_forward(x, W, nc, taps)
    if name.startswith("yolov6"):
        return yolov6_forward(x, W, name[-1], nc, taps)
    if name.startswith("yolov5"):
        return yolov5_forward(x, W, name[-1], nc)
    if name.startswith("yolov10"):
        return yolov10_forward(x, W, name[len("yolov10"):], nc, taps)
    if name == "yolov9c":
        return yolov9c_forward(x, W, nc, taps)
    if name.startswith("yolov9"):
        return yolov9t_forward(x, W, nc, taps)
    if name.startswith("yolov8"):
        return yolov8_forward(x, W, name[-1], nc, taps)
    raise ValueError(name)


# ------------------------------------------------------------------ YOLOv5
V5_SCALES = {"n": (0.33, 0.25), "s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}
V5_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def _c3(x, W, name, n, shortcut):
    y = _conv(x, W, f"{name}.cv1.conv")
    for i in range(n):
        t = _conv(_conv(y, W, f"{name}.m.{i}.cv1.conv"), W, f"{name}.m.{i}.cv2.conv")
        y = y + t if shortcut else t
    return _conv(torch.cat((y, _conv(x, W, f"{name}.cv2.conv")), 1), W, f"{name}.cv3.conv")


def yolov5_forward(x, W, scale="n", nc=80):
    depth = V5_SCALES[scale][0]
    dep = lambda n: max(round(n * depth), 1)
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        x = _conv(x, W, "model.0.conv", 2, 2)
        x = _conv(x, W, "model.1.conv", 2)
        x = _c3(x, W, "model.2", dep(3), True)
        x = _conv(x, W, "model.3.conv", 2)
        x4 = _c3(x, W, "model.4", dep(6), True)
        x = _conv(x4, W, "model.5.conv", 2)
        x6 = _c3(x, W, "model.6", dep(9), True)
        x = _conv(x6, W, "model.7.conv", 2)
        x = _c3(x, W, "model.8", dep(3), True)
        x = _sppf(x, W, "model.9")
        x10 = _conv(x, W, "model.10.conv")
        x = torch.cat((F.interpolate(x10, scale_factor=2, mode="nearest"), x6), 1)
        x = _c3(x, W, "model.13", dep(3), False)
        x14 = _conv(x, W, "model.14.conv")
        x = torch.cat((F.interpolate(x14, scale_factor=2, mode="nearest"), x4), 1)
        x17 = _c3(x, W, "model.17", dep(3), False)
        x = torch.cat((_conv(x17, W, "model.18.conv", 2), x14), 1)
        x20 = _c3(x, W, "model.20", dep(3), False)
        x = torch.cat((_conv(x20, W, "model.21.conv", 2), x10), 1)
        x23 = _c3(x, W, "model.23", dep(3), False)
        return _v5_decode((x17, x20, x23), W, "model.24.m.{}", nc, V5_ANCHORS, x17.shape[2] * 8)


def _v5_decode(feats, W, name_fmt, nc, anchors, H_in):
    """yolov5 models/yolo.py Detect.forward (inference branch): per level a 1x1 conv, sigmoid, then the grid / anchor decode."""
    no, z = nc + 5, []
    for i, f in enumerate(feats):
        y = _conv(f, W, name_fmt.format(i), act=None)
        bs, _, ny, nx = y.shape
        y = y.view(bs, 3, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous().sigmoid()
        stride = float(H_in // ny)
        yv, xv = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2)
        ag = torch.tensor(anchors[i], dtype=torch.float32).view(1, 3, 1, 1, 2)
        xy = (y[..., 0:2] * 2 - 0.5 + grid) * stride
        wh = (y[..., 2:4] * 2) ** 2 * ag
        z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(bs, -1, no))
    return torch.cat(z, 1).numpy()


# ------------------------------------------------------------------ YOLOv7-tiny
# WongKinYiu/yolov7 cfg/deploy/yolov7-tiny.yaml restated as its row table [from, module, args]; every Conv is conv + folded BN +
# LeakyReLU(0.1), MP = MaxPool2d(2, 2), SP(k) = MaxPool2d(k, 1, k // 2); the reference decodes the exported head as the v5 layout
# (yoloDetector.py:110-124).  Interpreted row by row exactly like upstream's parse_model / forward_once: y[i] = module(y[from]).
def _v7_elan(c, cout):
    return [(-1, "Conv", (c, 1, 1)), (-2, "Conv", (c, 1, 1)), (-1, "Conv", (c, 3, 1)), (-1, "Conv", (c, 3, 1)),
            ((-1, -2, -3, -4), "Concat", ()), (-1, "Conv", (cout, 1, 1))]


V7_TINY_ROWS = (
    [(-1, "Conv", (32, 3, 2)), (-1, "Conv", (64, 3, 2))] + _v7_elan(32, 64) +                                   # 0-7
    [(-1, "MP", ())] + _v7_elan(64, 128) + [(-1, "MP", ())] + _v7_elan(128, 256) + [(-1, "MP", ())] + _v7_elan(256, 512) +   # 8-28
    [(-1, "Conv", (256, 1, 1)), (-2, "Conv", (256, 1, 1)), (-1, "SP", (5,)), (-2, "SP", (9,)), (-3, "SP", (13,)),
     ((-1, -2, -3, -4), "Concat", ()), (-1, "Conv", (256, 1, 1)), ((-1, -7), "Concat", ()), (-1, "Conv", (256, 1, 1))] +      # 29-37
    [(-1, "Conv", (128, 1, 1)), (-1, "Up", ()), (21, "Conv", (128, 1, 1)), ((-1, -2), "Concat", ())] + _v7_elan(64, 128) +   # 38-47
    [(-1, "Conv", (64, 1, 1)), (-1, "Up", ()), (14, "Conv", (64, 1, 1)), ((-1, -2), "Concat", ())] + _v7_elan(32, 64) +      # 48-57
    [(-1, "Conv", (128, 3, 2)), ((-1, 47), "Concat", ())] + _v7_elan(64, 128) +                                               # 58-65
    [(-1, "Conv", (256, 3, 2)), ((-1, 37), "Concat", ())] + _v7_elan(128, 256) +                                              # 66-73
    [(57, "Conv", (128, 3, 1)), (65, "Conv", (256, 3, 1)), (73, "Conv", (512, 3, 1))])                                        # 74-76
V7_TINY_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def yolov7_tiny_forward(x, W, nc=80, taps=None):
    x = torch.as_tensor(x, dtype=torch.float32)
    H_in = x.shape[2]
    y = []
    with torch.no_grad():
        for i, (frm, mod, args) in enumerate(V7_TINY_ROWS):
            src = [x if not y else y[i + f if f < 0 else f] for f in (frm if isinstance(frm, tuple) else (frm,))]
            if mod == "Conv":
                c, k, s = args
                o = _conv(src[0], W, f"model.{i}.conv", s, act="leaky")
                assert o.shape[1] == c, (i, o.shape)
            elif mod == "MP":
                o = F.max_pool2d(src[0], 2, 2)
            elif mod == "SP":
                o = F.max_pool2d(src[0], args[0], 1, args[0] // 2)
            elif mod == "Up":
                o = F.interpolate(src[0], scale_factor=2, mode="nearest")
            else:
                o = torch.cat(src, 1)
            y.append(o)
        if taps is not None:
            taps.update(p3=y[74], p4=y[75], p5=y[76], sppcspc=y[37])
        return _v5_decode(y[74:77], W, "model.77.m.{}", nc, V7_TINY_ANCHORS, H_in)


# ------------------------------------------------------------------ UFLDv2
RESNET_DEPTHS = {"18": [2, 2, 2, 2], "34": [3, 4, 6, 3]}


def ufldv2_forward(x, W, backbone="18", num_grid_row=200, num_cls_row=72, num_grid_col=100, num_cls_col=81,
                   num_lanes=4, taps=None, fc_norm=True):
    """Returns [loc_row, loc_col, exist_row, exist_col] (model_culane.py:56-59)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        x = _conv(x, W, "model.conv1", 2, 3, "relu")
        x = F.max_pool2d(x, 3, 2, 1)
        cin = 64
        for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], RESNET_DEPTHS[backbone])):
            for bi in range(nblk):
                s = 2 if (li > 0 and bi == 0) else 1
                name = f"model.layer{li + 1}.{bi}"
                idt = x
                if s != 1 or cin != planes:
                    idt = _conv(x, W, f"{name}.downsample.0", s, 0, None)
                t = _conv(x, W, f"{name}.conv1", s, 1, "relu")
                x = F.relu(_conv(t, W, f"{name}.conv2", 1, 1, None) + idt)
                cin = planes
        if taps is not None:
            taps["layer4"] = x
        fea = _conv(x, W, "pool", 1, 0, None)
        fea = fea.reshape(fea.shape[0], -1)                                   # (C,H,W) flatten, model_culane.py:53
        if taps is not None:
            taps["fea"] = fea
        if fc_norm:                                                           # LayerNorm | Identity (model_culane.py:34; tusimple: fc_norm=False)
            fea = F.layer_norm(fea, (fea.shape[1],), _t(W, "cls.0.weight"), _t(W, "cls.0.bias"), 1e-5)
        h = F.relu(F.linear(fea, _t(W, "cls.1.weight"), _t(W, "cls.1.bias")))
        out = F.linear(h, _t(W, "cls.3.weight"), _t(W, "cls.3.bias"))
        d1 = num_grid_row * num_cls_row * num_lanes
        d2 = num_grid_col * num_cls_col * num_lanes
        d3 = 2 * num_cls_row * num_lanes
        d4 = 2 * num_cls_col * num_lanes
        N = out.shape[0]
        return [out[:, :d1].reshape(N, num_grid_row, num_cls_row, num_lanes).numpy(),
                out[:, d1:d1 + d2].reshape(N, num_grid_col, num_cls_col, num_lanes).numpy(),
                out[:, d1 + d2:d1 + d2 + d3].reshape(N, 2, num_cls_row, num_lanes).numpy(),
                out[:, -d4:].reshape(N, 2, num_cls_col, num_lanes).numpy()]


def ufld_v1_forward(x, W, backbone="18", griding_num=100, cls_num_per_lane=56, num_lanes=4):
    """UFLD (v1) parsingNet (exportLib/ultrafastLane/model.py:19-89): ResNet trunk, `pool` 1x1
    conv to 8 channels, (C,H,W) flatten, Linear-ReLU-Linear, view (N, G+1, K, L) -- the tensor
    ultrafastLaneDetector.py:99-109 consumes."""
    x = torch.as_tensor(x, dtype=torch.float32)
    with torch.no_grad():
        x = _conv(x, W, "model.conv1", 2, 3, "relu")
        x = F.max_pool2d(x, 3, 2, 1)
        cin = 64
        for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], RESNET_DEPTHS[backbone])):
            for bi in range(nblk):
                s = 2 if (li > 0 and bi == 0) else 1
                name = f"model.layer{li + 1}.{bi}"
                idt = x
                if s != 1 or cin != planes:
                    idt = _conv(x, W, f"{name}.downsample.0", s, 0, None)
                t = _conv(x, W, f"{name}.conv1", s, 1, "relu")
                x = F.relu(_conv(t, W, f"{name}.conv2", 1, 1, None) + idt)
                cin = planes
        fea = _conv(x, W, "pool", 1, 0, None)
        fea = fea.reshape(fea.shape[0], -1)
        h = F.relu(F.linear(fea, _t(W, "cls.0.weight"), _t(W, "cls.0.bias")))
        out = F.linear(h, _t(W, "cls.2.weight"), _t(W, "cls.2.bias"))
        return out.reshape(out.shape[0], griding_num + 1, cls_num_per_lane, num_lanes).numpy()
