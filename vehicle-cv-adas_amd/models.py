"""Network graphs for HipEngine + the `.hipm` model container.

The reference ships no weights and (for YOLO) no architecture: it loads whatever `.onnx`/`.trt`
the user exported (coreEngine.py:120-186; I/O contract yoloDetector.py:110-133).  HipEngine instead
loads a self-describing container ("ADASHIP1") holding a flat list of NHWC ops over channel-sliced
buffer views plus fp32 weights.  This module builds those graphs:

  * YOLOv8 n/s/m/l  (ultralytics 8.1.x yolov8.yaml; README.md:56)       -> output (1, 4+nc, 8400)
  * YOLOv5 n/s      (yolov5 v6.2 yolov5n.yaml; README.md:53)            -> output (1, 25200, 5+nc)
  * UFLDv2 CULane ResNet-18/34 (exportLib/ultrafastLaneV2/model_culane.py:7-63,
    backbone.py:14-58, configs/culane_res18.py)                        -> 4 outputs (a1 in SURVEY 8a)

Concat / split / residual never copy: producers write straight into channel offsets of the consumer's
concat buffer ("view" = buffer id + channel offset + channel count).  BatchNorm is pre-folded into conv
weight + bias.  Weights come from a `wsrc(name, shape, kind)` callable; the default is a seeded
synthetic generator (He-style, Detect biases as upstream: cls log(5/nc/(640/s)^2), box 1.0).
"""
import math
import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"ADASHIP1"
(OP_INPUT, OP_CONV, OP_MAXPOOL, OP_UPSAMPLE2, OP_DETECT_V8, OP_DETECT_V5, OP_LAYERNORM, OP_DWCONV, OP_ATTENTION, OP_AVGPOOL, OP_DEPTH2SPACE,
 OP_DETECT_V6, OP_SE_GATE, OP_SCALE, OP_WSUM, OP_SHUFFLE) = range(16)
ACT_NONE, ACT_SILU, ACT_RELU, ACT_LEAKY = 0, 1, 2, 3      # ACT_LEAKY: LeakyReLU(0.1) (YOLOv7)
ACT_HSWISH, ACT_HSIGMOID, ACT_RELU6 = 4, 5, 6    # torch.nn.Hardswish / Hardsigmoid / ReLU6: ELEMENT-WISE layers only (Graph.act, the gate of Graph.se), never a conv epilogue
RES_NONE, RES_AFTER_ACT, RES_BEFORE_ACT = 0, 1, 2
BUF_F32 = 1
BUF_ALIAS = 2      # flags bit 1: this buffer is another view of buffer (flags >> 8) - same bytes, different (h, w, c)

HDR_FMT = "<8sIIIIIIIIQQd64s"
BUF_FMT = "<IIII"
OP_FMT = "<II8i8i8iiiiIIIIIIiiIIIQQQQd8f48s"
OUT_FMT = "<III4I32s4x"
HDR_SIZE, BUF_SIZE, OP_SIZE, OUT_SIZE = (struct.calcsize(f) for f in (HDR_FMT, BUF_FMT, OP_FMT, OUT_FMT))


# Seeded synthetic weights: He-style init (std = gain * sqrt(2 / fan_in)) with a per-graph gain.
#
# A random SiLU network has no stable O(1) operating point: above a critical gain (~1.11 for YOLOv8n/s, lower for the deeper
# scales, whose chains of residual Bottlenecks add variance) the activations grow with depth AND the net is chaotic -- a
# perturbation (e.g. one 16-bit rounding) is amplified ~1.1x per layer, so a whole-network 16-bit comparison measures the weights'
# chaos, not the kernels (round 2: gain 1.15 -> YOLOv8n fp16 rel-L2 6e-4 after one layer, 1.6e-2 at the head; class probabilities
# off by 0.25 on YOLOv8s).  Below it the activations settle at the bias-driven level (rms ~0.04) and a perturbation keeps its
# relative size through the depth (measured with oracle/nets.py EMULATE="fp16", tools/synth_snr.py: logit error / logit
# signal 3.4e-3 at gain 1.15 vs 7e-4 at 1.08, box error 0.12 px rms vs 6e-4 px).  The synthetic nets are therefore built just
# BELOW the critical gain of their scale -- not far below: there the input-dependent signal itself decays with depth while the
# rounding noise of the last layers does not (tools/synth_snr.py: anchor-to-anchor spread of the best class logit over its
# fp16 error = 400-1500 just below the critical gain, 77-200 just above).  A trained checkpoint needs none of this (DictWeights).
SILU_GAIN = 1.10                # YOLOv8 n/s: AT the critical gain (n ~1.11, s ~1.10).  Further below it the score field gets so smooth that
                                # neighbouring candidates tie within the 16-bit error and the reference's score-ordered NMS picks other
                                # survivors (tools/synth_flips.py, YOLOv8s: identical survivor sets 25 % at 1.08, 92 % at 1.09, 100 % at 1.10);
                                # 0.02 above it the rounding error itself is amplified 3-10x
V5_SILU_GAIN = 1.15             # YOLOv5 (C3 blocks; kept at the round-1 value: its deeper scales are chaotic there, bf16 head rel-L2 6e-2,
                                # and no better at 1.0)
RELU_RES_GAIN = 0.8             # ResNet lane nets: ReLU + residual adds double the variance; flat drift at 0.8 (9e-4 rel-L2 fp16)
SYNTH_GAINS = {"yolov6n": 0.95, "yolov6s": 0.95,                      # plain ReLU 3x3 stacks (no residuals): He gain 1 holds the variance; 0.95 decays gently
               "yolov9c": 1.06,
               "yolov9s": 1.06,                                       # 1.12 (yolov9t's) has a runaway mode on small inputs (96x128: rms 1e5 at P5)
               "yolov7-tiny": 1.0,                                    # LeakyReLU: piecewise linear, no chaos (fp16 rel-L2 1.3e-3 at any gain); 1.0 keeps rms ~0.4
               "yolov9t": 1.12,                                       # critical between 1.16 and 1.22 (activations explode there)
               "yolov10s": 1.0,
               "yolov10n": 1.05,                                      # critical ~1.09 (1.08 already drifts: fp16 rel-L2 1.9e-3 at P5, boxes 0.3 px)
               "efficientdet-d0": 1.0,                                # = EFFDET_GAIN
               "yolov8m": 0.99, "yolov8l": 0.96, "yolov8x": 0.98}     # deeper Bottleneck chains: critical gain ~1.03 (m), ~0.97 (l), ~1.0 (x)


def synth_gain(name):
    """Gain of the seeded synthetic weights for graph `name` (a trained checkpoint needs none of this)."""
    if name in SYNTH_GAINS:
        return SYNTH_GAINS[name]
    if name.startswith("ufld"):
        return RELU_RES_GAIN
    return V5_SILU_GAIN if name.startswith("yolov5") else SILU_GAIN


class View:
    """A channel slice [coff, coff+c) of NHWC buffer `buf`."""
    __slots__ = ("buf", "coff", "c", "h", "w")

    def __init__(self, buf, coff, c, h, w):
        self.buf, self.coff, self.c, self.h, self.w = buf, coff, c, h, w

    def slice(self, off, c):
        assert 0 <= off and off + c <= self.c
        return View(self.buf, self.coff + off, c, self.h, self.w)


class SynthWeights:
    """Seeded synthetic parameters.  kind: 'conv' (OIHW), 'bias', 'linear' (out,in), 'ln_w', 'ln_b'."""

    def __init__(self, seed=0, gain=1.0):
        self.rng = np.random.default_rng(seed)
        self.gain = gain
        self.store = OrderedDict()

    def __call__(self, name, shape, kind, fill=None):
        if name in self.store:
            return self.store[name]
        if fill is not None:
            a = np.full(shape, fill, np.float32)
        elif kind in ("conv", "linear"):
            fan_in = int(np.prod(shape[1:]))
            a = (self.rng.standard_normal(shape) * self.gain * math.sqrt(2.0 / fan_in)).astype(np.float32)
        elif kind == "bias":
            a = (self.rng.standard_normal(shape) * 0.05).astype(np.float32)
        elif kind == "ln_w":
            a = (1.0 + 0.05 * self.rng.standard_normal(shape)).astype(np.float32)
        elif kind == "ln_b":
            a = (0.05 * self.rng.standard_normal(shape)).astype(np.float32)
        else:
            raise ValueError(kind)
        self.store[name] = a
        return a


class DictWeights:
    """Weights supplied by name (e.g. converted from a real checkpoint)."""

    def __init__(self, d):
        self.store = d

    def __call__(self, name, shape, kind, fill=None):
        a = np.asarray(self.store[name], np.float32)
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        return a


class Graph:
    def __init__(self, name, in_c, in_h, in_w, wsrc):
        self.name, self.in_c, self.in_h, self.in_w = name, in_c, in_h, in_w
        self.w = wsrc
        self.bufs, self.ops, self.outs = [], [], []
        self.blob = bytearray()
        self.flops = 0.0
        self.n_convs = 0
        self.n_params = 0
        self.io_half = False   # the source model's graph I/O is float16 (header flag; HipEngine.engine_dtype follows it)

    # ---- buffers / views
    def buf(self, h, w, c, f32=False):
        self.bufs.append((h, w, c, BUF_F32 if f32 else 0))
        return View(len(self.bufs) - 1, 0, c, h, w)

    def alias(self, view, h, w, c):
        """The same memory as `view`'s buffer, re-declared as an (h, w, c) tensor (torch .view(-1, N), model_culane.py:53)."""
        hh, ww, cc, fl = self.bufs[view.buf]
        assert view.coff == 0 and view.c == cc and hh * ww * cc == h * w * c
        self.bufs.append((h, w, c, (fl & BUF_F32) | BUF_ALIAS | (view.buf << 8)))
        return View(len(self.bufs) - 1, 0, c, h, w)

    def _blob(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        off = len(self.blob)
        self.blob += arr.tobytes()
        pad = (-len(self.blob)) % 64
        self.blob += b"\0" * pad
        return off, arr.size

    def _op(self, typ, ins, out, kh=0, kw=0, stride=1, pad=0, act=0, res_mode=0, res=None, w=(0, 0), b=(0, 0),
            flops=0.0, params=(), name=""):
        ins = list(ins)
        rec = dict(type=typ, ins=ins, out=out, kh=kh, kw=kw, stride=stride, pad=pad, act=act, res_mode=res_mode, res=res,
                   w=w, b=b, flops=float(flops), params=list(params), name=name)
        self.ops.append(rec)
        self.flops += float(flops)
        return rec

    # ---- ops
    def input(self):
        """NCHW fp32 network input -> NHWC with channels zero-padded to 8."""
        v = self.buf(self.in_h, self.in_w, 8)
        self._op(OP_INPUT, [], v, name="input")
        return View(v.buf, 0, 8, v.h, v.w), self.in_c

    def conv(self, x, cout, k, s, name, act=ACT_SILU, out=None, res=None, res_mode=RES_NONE, true_cin=None, bias=True,
             wname=None, bname=None, pad=None, f32_out=False, bias_fill=None, wkind="conv", weight=None, bias_arr=None):
        """x: View (its .c may be zero-padded beyond true_cin).  Weight 'name.weight' is OIHW with I=true_cin.
        weight / bias_arr: explicit arrays (a re-arranged parameter the caller owns and counts, e.g. deconv2x2)."""
        cin_true = true_cin if true_cin is not None else x.c
        p = (k // 2) if pad is None else pad
        ho = (x.h + 2 * p - k) // s + 1
        wo = (x.w + 2 * p - k) // s + 1
        if out is None:
            out = self.buf(ho, wo, cout, f32=f32_out)
        assert (out.h, out.w, out.c) == (ho, wo, cout), (name, (out.h, out.w, out.c), (ho, wo, cout))
        wshape = (cout, cin_true, k, k) if wkind == "conv" else (cout, cin_true)
        W = self.w(wname or name + ".weight", wshape, wkind) if weight is None else np.asarray(weight, np.float32).reshape(wshape)
        W4 = W.reshape(cout, cin_true, k, k)
        if bias_arr is not None:
            B = np.asarray(bias_arr, np.float32).reshape(cout)
        else:
            B = self.w(bname or name + ".bias", (cout,), "bias", fill=bias_fill) if bias else np.zeros(cout, np.float32)
        if k == 1 and cin_true == x.c:
            ohwi = W4.reshape(cout, cin_true)                     # 1x1 / linear: OIHW == OHWI
        else:
            ohwi = np.zeros((cout, k, k, x.c), np.float32)       # channel-padded OHWI
            ohwi[..., :cin_true] = W4.transpose(0, 2, 3, 1)
        woff = self._blob(ohwi)
        boff = self._blob(B)
        fl = 2.0 * ho * wo * cout * cin_true * k * k
        self._op(OP_CONV, [x], out, kh=k, kw=k, stride=s, pad=p, act=act, res_mode=res_mode, res=res, w=woff, b=boff,
                 flops=fl, name=name)
        self.n_convs += 1
        if weight is None:
            self.n_params += W.size + (B.size if bias else 0)
        return out

    def conv_siblings(self, x, specs, k, s, out, act=ACT_SILU, draw=None):
        """Convs that read the SAME tensor with the same kernel / stride / activation and write ADJACENT channel slices of one buffer
        (C3 / RepCSP cv1 | cv2, ELAN's two 1x1 branches) as ONE conv with the weight matrices stacked: the input is read once and a launch
        goes away; every output channel keeps its own arithmetic (same K order), so the values are those of the separate convs.
        specs: [(cout, name)] in the channel order of `out` (a view of sum(cout) channels).  Weights keep their own names."""
        couts = [c for c, _ in specs]
        assert out.c == sum(couts)
        Ws, Bs = [None] * len(specs), [None] * len(specs)
        for i in (draw if draw is not None else range(len(specs))):      # parameter request order = the module order of the separate convs
            c, n = specs[i]                                               # (a seeded SynthWeights source draws in request order)
            Ws[i] = self.w(n + ".weight", (c, x.c, k, k), "conv")
            Bs[i] = self.w(n + ".bias", (c,), "bias")
        name = specs[0][1] + "".join("+" + n.rsplit(".", 2)[-2] + "." + n.rsplit(".", 1)[-1] if n.count(".") >= 2 else "+" + n for _, n in specs[1:])
        assert len(name) < 48, name
        self.conv(x, out.c, k, s, name, act=act, out=out, weight=np.concatenate(Ws, 0), bias_arr=np.concatenate(Bs, 0))
        self.n_params += sum(w.size for w in Ws) + sum(b.size for b in Bs)
        self.n_convs += len(specs) - 1
        off, views = 0, []
        for c in couts:
            views.append(out.slice(off, c))
            off += c
        return views

    def deconv2x2(self, x, cout, name, out=None):
        """nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2, bias=True) (YOLOv6 Transpose): weight 'name.weight' in torch's (cin, cout, 2, 2)
        layout.  out[2y + dy, 2x + dx] = W[:, :, dy, dx]^T x[y, x] + b: a 1x1 conv to 4 * cout channels (rows ordered (dy, dx, cout): MFMA work,
        conv_pw) followed by a depth-to-space move."""
        cin = x.c
        Wt = self.w(name + ".weight", (cin, cout, 2, 2), "conv")
        B = self.w(name + ".bias", (cout,), "bias")
        t = self.conv(x, 4 * cout, 1, 1, name, act=ACT_NONE, weight=Wt.transpose(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1), bias_arr=np.tile(B, 4))
        self.n_params += Wt.size + B.size
        if out is None:
            out = self.buf(2 * x.h, 2 * x.w, cout)
        assert (out.h, out.w, out.c) == (2 * x.h, 2 * x.w, cout)
        self._op(OP_DEPTH2SPACE, [t], out, name=name + ".d2s")
        return out

    def dwconv(self, x, k, s, name, act=ACT_SILU, out=None, res=None, weight=None, bias=None):
        """Depth-wise k x k conv (groups = channels, pad k // 2), BatchNorm folded: weight 'name.weight' (C,1,k,k), 'name.bias' (C,).
        res: added after the activation (x + block(x)).  weight / bias: explicit arrays (a channel slice of a shared parameter)."""
        c, p = x.c, k // 2
        ho, wo = (x.h + 2 * p - k) // s + 1, (x.w + 2 * p - k) // s + 1
        if out is None:
            out = self.buf(ho, wo, c)
        assert (out.h, out.w, out.c) == (ho, wo, c), (name, (out.h, out.w, out.c), (ho, wo, c))
        W = self.w(name + ".weight", (c, 1, k, k), "conv") if weight is None else np.asarray(weight, np.float32)
        B = self.w(name + ".bias", (c,), "bias") if bias is None else np.asarray(bias, np.float32)
        assert W.shape == (c, 1, k, k) and B.shape == (c,)
        woff, boff = self._blob(W.reshape(c, k * k)), self._blob(B)
        self._op(OP_DWCONV, [x], out, kh=k, kw=k, stride=s, pad=p, act=act, res_mode=RES_AFTER_ACT if res is not None else RES_NONE, res=res,
                 w=woff, b=boff, flops=2.0 * ho * wo * c * k * k, name=name)
        self.n_convs += 1
        if weight is None:
            self.n_params += W.size + B.size
        return out

    def attention(self, qkv, num_heads, key_dim, head_dim, name, out=None):
        """Softmax attention over the H*W tokens of `qkv` (channels per head: key_dim q, key_dim k, head_dim v): ultralytics
        Attention.forward without its qkv / proj / pe convolutions."""
        assert qkv.c == num_heads * (2 * key_dim + head_dim)
        if out is None:
            out = self.buf(qkv.h, qkv.w, num_heads * head_dim)
        n = qkv.h * qkv.w
        self._op(OP_ATTENTION, [qkv], out, params=[num_heads, key_dim, head_dim, float(key_dim) ** -0.5],
                 flops=2.0 * num_heads * n * n * (key_dim + head_dim), name=name)
        return out

    def maxpool(self, x, k, s, p, out=None, name="maxpool"):
        ho = (x.h + 2 * p - k) // s + 1
        wo = (x.w + 2 * p - k) // s + 1
        if out is None:
            out = self.buf(ho, wo, x.c)
        assert (out.h, out.w, out.c) == (ho, wo, x.c)
        self._op(OP_MAXPOOL, [x], out, kh=k, kw=k, stride=s, pad=p, name=name)
        return out

    def avgpool(self, x, k, s, p, out=None, name="avgpool"):
        """F.avg_pool2d(x, k, s, p, ceil_mode=False, count_include_pad=True)."""
        ho = (x.h + 2 * p - k) // s + 1
        wo = (x.w + 2 * p - k) // s + 1
        if out is None:
            out = self.buf(ho, wo, x.c)
        assert (out.h, out.w, out.c) == (ho, wo, x.c)
        self._op(OP_AVGPOOL, [x], out, kh=k, kw=k, stride=s, pad=p, name=name)
        return out

    def upsample2(self, x, out=None, name="upsample"):
        if out is None:
            out = self.buf(x.h * 2, x.w * 2, x.c)
        assert (out.h, out.w, out.c) == (x.h * 2, x.w * 2, x.c)
        self._op(OP_UPSAMPLE2, [x], out, name=name)
        return out

    def se(self, x, cr, name, out=None, hidden_act=ACT_SILU, gate_act=ACT_NONE):
        """Squeeze-and-excitation (EfficientNet MBConv): x * sigmoid(W2 silu(W1 mean_hw(x) + b1) + b2), squeeze width cr; hidden_act =
        ACT_RELU and gate_act = ACT_HSIGMOID give the MobileNetV3 / PP-LCNet form x * hardsigmoid(W2 relu(W1 mean + b1) + b2).  Parameters
        'name.reduce.{weight,bias}' (cr, C, 1, 1) and 'name.expand.{weight,bias}' (C, cr, 1, 1).  Two launches: the gate (one fp32 value
        per frame and channel, fp32 arithmetic in every precision) and the channel scale."""
        c = x.c
        W1, b1 = self.w(name + ".reduce.weight", (cr, c, 1, 1), "conv"), self.w(name + ".reduce.bias", (cr,), "bias")
        W2, b2 = self.w(name + ".expand.weight", (c, cr, 1, 1), "conv"), self.w(name + ".expand.bias", (c,), "bias")
        gate = self.buf(1, 1, c, f32=True)
        # maps of >= 1024 pixels: 16 pixel ranges per frame are summed by their own launch into this scratch (res slot of the op), so
        # that a batch fills the chip; the gate kernel then adds the ranges in order (results do not depend on the split being used:
        # the sums are the same stripes in another grouping -- fp32, ~1e-7 apart)
        scratch = self.buf(1, 1, 16 * c, f32=True) if x.h * x.w >= 1024 and c <= 2048 else None
        self._op(OP_SE_GATE, [x], gate, w=self._blob(np.concatenate([W1.ravel(), b1])), b=self._blob(np.concatenate([W2.ravel(), b2])),
                 res=scratch, params=[cr, 0 if hidden_act == ACT_SILU else hidden_act, gate_act], flops=x.h * x.w * c + 4.0 * c * cr, name=name + ".gate")
        if out is None:
            out = self.buf(x.h, x.w, c)
        assert (out.h, out.w, out.c) == (x.h, x.w, c)
        self._op(OP_SCALE, [x, gate], out, flops=float(x.h * x.w * c), name=name + ".scale")
        self.n_params += W1.size + b1.size + W2.size + b2.size
        return out

    def wsum(self, ins, weights, name, act=ACT_SILU, out=None):
        """act(sum_i weights[i] * ins[i]) over 2-3 maps of one width; an input of half the output's resolution is read through a nearest
        2x upsample (BiFPN top-down nodes).  The output resolution is that of the largest input."""
        assert 1 <= len(ins) <= 3 and len(weights) == len(ins)
        assert act in (ACT_NONE, ACT_SILU, ACT_RELU, ACT_LEAKY, ACT_HSWISH, ACT_HSIGMOID, ACT_RELU6), (name, act)   # the kernel applies exactly these (fuse_ops.hip wsum_kernel)
        h, w, c = max(v.h for v in ins), max(v.w for v in ins), ins[0].c
        for v in ins:
            assert v.c == c and ((v.h, v.w) == (h, w) or (2 * v.h, 2 * v.w) == (h, w)), (name, (v.h, v.w, v.c), (h, w, c))
        if out is None:
            out = self.buf(h, w, c)
        assert (out.h, out.w, out.c) == (h, w, c)
        self._op(OP_WSUM, list(ins), out, act=act, params=[float(np.float32(x)) for x in weights], flops=2.0 * len(ins) * h * w * c, name=name)
        return out

    def act(self, x, act, name, out=None):
        """A stand-alone activation layer (one-input weighted sum): how hard-swish / hard-sigmoid networks run -- the convolution in front
        keeps ACT_NONE and this layer applies the function (the conv epilogues carry SiLU / ReLU / LeakyReLU only)."""
        return self.wsum([x], [1.0], name, act=act, out=out)

    def shuffle(self, x, groups, name, out=None):
        """torch channel_shuffle(x, groups) (ShuffleNetV2 units): out channel j * groups + i = in channel i * (C / groups) + j."""
        assert x.c % groups == 0 and x.c % 8 == 0
        if out is None:
            out = self.buf(x.h, x.w, x.c)
        assert (out.h, out.w, out.c) == (x.h, x.w, x.c)
        self._op(OP_SHUFFLE, [x], out, params=[groups], name=name)
        return out

    def output(self, view, offset, dims, name):
        self.outs.append((view.buf, int(offset), list(dims), name))

    # ---- serialisation
    def tobytes(self):
        parts = []
        nb, no, nout = len(self.bufs), len(self.ops), len(self.outs)
        woff = HDR_SIZE + nb * BUF_SIZE + no * OP_SIZE + nout * OUT_SIZE
        woff += (-woff) % 256
        parts.append(struct.pack(HDR_FMT, MAGIC, 1, nb, no, nout, self.in_c, self.in_h, self.in_w, 8 | (int(self.io_half) << 16), woff, len(self.blob),
                                 self.flops, self.name.encode()[:63]))
        for h, w, c, fl in self.bufs:
            parts.append(struct.pack(BUF_FMT, h, w, c, fl))
        for r in self.ops:
            ins = r["ins"]
            ib = [v.buf for v in ins] + [-1] * (8 - len(ins))
            io = [v.coff for v in ins] + [0] * (8 - len(ins))
            ic = [v.c for v in ins] + [0] * (8 - len(ins))
            res = r["res"]
            prm = (r["params"] + [0.0] * 8)[:8]
            parts.append(struct.pack(OP_FMT, r["type"], len(ins), *ib, *io, *ic, r["out"].buf, r["out"].coff, r["out"].c,
                                     r["kh"], r["kw"], r["stride"], r["pad"], r["act"], r["res_mode"],
                                     res.buf if res else -1, res.coff if res else 0, 0, 0, 0,
                                     r["w"][0], r["w"][1], r["b"][0], r["b"][1], r["flops"], *prm,
                                     r["name"].encode()[:47]))
        for buf, off, dims, name in self.outs:
            d = (list(dims) + [1] * 4)[:4]
            parts.append(struct.pack(OUT_FMT, buf, off, len(dims), *d, name.encode()[:31]))
        head = b"".join(parts)
        head += b"\0" * (woff - len(head))
        return head + bytes(self.blob)

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.tobytes())
        return path


# =====================================================================================
# YOLOv8  (ultralytics yolov8.yaml)
# =====================================================================================
V8_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.0, 1.0, 512),
             "x": (1.0, 1.25, 512)}


def _hw(imgsz):
    """imgsz: one int (square) or (H, W); both sides must be multiples of the largest stride (32)."""
    H, W = (int(imgsz), int(imgsz)) if np.isscalar(imgsz) else (int(imgsz[0]), int(imgsz[1]))
    if H <= 0 or W <= 0 or H % 32 or W % 32:
        raise ValueError("YOLO input size must be positive multiples of 32, got %sx%s" % (H, W))
    return H, W


def _mk(c, width, max_ch):
    return int(math.ceil(min(c, max_ch) * width / 8) * 8)


def _c2f(g, x, c2, n, shortcut, name, out=None):
    """C2f: cv1 1x1 -> split; n Bottlenecks chained on the last chunk; cv2 1x1 over the concat."""
    c = c2 // 2
    cat = g.buf(x.h, x.w, (2 + n) * c)
    g.conv(x, 2 * c, 1, 1, f"{name}.cv1.conv", out=cat.slice(0, 2 * c))
    for i in range(n):
        src = cat.slice((1 + i) * c, c)
        t = g.conv(src, c, 3, 1, f"{name}.m.{i}.cv1.conv")
        g.conv(t, c, 3, 1, f"{name}.m.{i}.cv2.conv", out=cat.slice((2 + i) * c, c),
               res=src if shortcut else None, res_mode=RES_AFTER_ACT if shortcut else RES_NONE)
    return g.conv(cat, c2, 1, 1, f"{name}.cv2.conv", out=out)


def _sppf(g, x, c2, name, out=None):
    c_ = x.c // 2
    cat = g.buf(x.h, x.w, 4 * c_)
    g.conv(x, c_, 1, 1, f"{name}.cv1.conv", out=cat.slice(0, c_))
    for i in range(3):
        g.maxpool(cat.slice(i * c_, c_), 5, 1, 2, out=cat.slice((i + 1) * c_, c_), name=f"{name}.m{i}")
    return g.conv(cat, c2, 1, 1, f"{name}.cv2.conv", out=out)


def yolov8(scale="n", nc=80, imgsz=640, wsrc=None, seed=0):
    depth, width, max_ch = V8_SCALES[scale]
    wsrc = wsrc or SynthWeights(seed, gain=synth_gain(f"yolov8{scale}"))
    H, W = _hw(imgsz)
    g = Graph(f"yolov8{scale}", 3, H, W, wsrc)
    ch = lambda c: _mk(c, width, max_ch)
    dep = lambda n: max(round(n * depth), 1)
    c1, c2, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    x, cin = g.input()
    # concat buffers of the neck (producers write into them directly)
    cat11 = g.buf(H // 16, W // 16, c5 + c4)   # [up(9), 6]
    cat14 = g.buf(H // 8, W // 8, c4 + c3)     # [up(12), 4]
    cat17 = g.buf(H // 16, W // 16, c3 + c4)   # [16, 12]
    cat20 = g.buf(H // 32, W // 32, c4 + c5)   # [19, 9]
    x = g.conv(x, c1, 3, 2, "model.0.conv", true_cin=cin)
    x = g.conv(x, c2, 3, 2, "model.1.conv")
    x = _c2f(g, x, c2, dep(3), True, "model.2")
    x = g.conv(x, c3, 3, 2, "model.3.conv")
    p3b = _c2f(g, x, c3, dep(6), True, "model.4", out=cat14.slice(c4, c3))
    x = g.conv(p3b, c4, 3, 2, "model.5.conv")
    p4b = _c2f(g, x, c4, dep(6), True, "model.6", out=cat11.slice(c5, c4))
    x = g.conv(p4b, c5, 3, 2, "model.7.conv")
    x = _c2f(g, x, c5, dep(3), True, "model.8")
    p5b = _sppf(g, x, c5, "model.9", out=cat20.slice(c4, c5))
    g.upsample2(p5b, out=cat11.slice(0, c5), name="model.10")
    n12 = _c2f(g, cat11, c4, dep(3), False, "model.12", out=cat17.slice(c3, c4))
    g.upsample2(n12, out=cat14.slice(0, c4), name="model.13")
    p3 = _c2f(g, cat14, c3, dep(3), False, "model.15")
    g.conv(p3, c3, 3, 2, "model.16.conv", out=cat17.slice(0, c3))
    p4 = _c2f(g, cat17, c4, dep(3), False, "model.18")
    g.conv(p4, c4, 3, 2, "model.19.conv", out=cat20.slice(0, c4))
    p5 = _c2f(g, cat20, c5, dep(3), False, "model.21")
    # Detect
    feats = [p3, p4, p5]
    cb = max(16, feats[0].c // 4, 64)
    cc = max(feats[0].c, min(nc, 100))
    ins, strides = [], []
    for i, f in enumerate(feats):
        s = H // f.h
        strides.append(s)
        b = g.conv(f, cb, 3, 1, f"model.22.cv2.{i}.0.conv")
        b = g.conv(b, cb, 3, 1, f"model.22.cv2.{i}.1.conv")
        b = g.conv(b, 64, 1, 1, f"model.22.cv2.{i}.2", act=ACT_NONE, f32_out=True, bias_fill=1.0)
        c = g.conv(f, cc, 3, 1, f"model.22.cv3.{i}.0.conv")
        c = g.conv(c, cc, 3, 1, f"model.22.cv3.{i}.1.conv")
        c = g.conv(c, nc, 1, 1, f"model.22.cv3.{i}.2", act=ACT_NONE, f32_out=True,
                   bias_fill=math.log(5 / nc / (640 / s) ** 2))      # upstream Detect.bias_init: the constant 640
        ins += [b, c]
    A = sum(f.h * f.w for f in feats)
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="model.22.decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    g.meta = dict(kind="yolov8", nc=nc, anchors=A, strides=strides)
    return g


# =====================================================================================
# YOLOv10 (THU-MIG yolov10n.yaml / ultralytics 8.1 fork: SCDown, PSA, C2fCIB, v10Detect) -- the reference's shipped default detector
# (demo.py:24-30 yolov10n-coco_fp16.trt, ObjectModelType.YOLOV10).  yoloDetector.py:114,121 transposes and decodes its output as a
# v8-layout (1, 4+nc, A) tensor [cx, cy, w, h, class probabilities]: the graph ends in the v8 decode over the ONE-TO-ONE head
# (the branch v10 deploys; same Detect arithmetic as v8: DFL expectation, dist2bbox, sigmoid).
# =====================================================================================
V10_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024)}      # yolov10s.yaml: row 8 is a C2fCIB(lk) instead of n's C2f


def _scdown(g, x, c2, k, s, name, out=None):
    """SCDown: 1x1 Conv (SiLU) then depth-wise k x k stride-s conv without activation."""
    t = g.conv(x, c2, 1, 1, f"{name}.cv1.conv")
    return g.dwconv(t, k, s, f"{name}.cv2.conv", act=ACT_NONE, out=out)


def _psa_block(g, x, name, out=None):
    """PSA(c1, c1, e=0.5): cv1 -> (a, b); b = b + Attention(b); b = b + FFN(b); cv2(cat(a, b)).
    Attention(dim=c, num_heads=c // 64, attn_ratio=0.5): qkv 1x1 (no act) -> per head q, k (key_dim = head_dim / 2) and v (head_dim);
    softmax(q^T k / sqrt(key_dim)) applied to v, plus pe (depth-wise 3x3, no act) of v; proj 1x1 (no act)."""
    c = x.c // 2
    nh = c // 64
    hd = c // nh
    kd = hd // 2
    ab = g.buf(x.h, x.w, 2 * c)
    g.conv(x, 2 * c, 1, 1, f"{name}.cv1.conv", out=ab)
    b = ab.slice(c, c)
    qkv = g.conv(b, c + 2 * nh * kd, 1, 1, f"{name}.attn.qkv.conv", act=ACT_NONE)
    att = g.attention(qkv, nh, kd, hd, f"{name}.attn.softmax")
    # + pe(v.reshape(B, C, H, W)): output channel h * hd + d is v of head h, i.e. qkv channel h * (2 kd + hd) + 2 kd + d
    pw = g.w(f"{name}.attn.pe.conv.weight", (c, 1, 3, 3), "conv")
    pb = g.w(f"{name}.attn.pe.conv.bias", (c,), "bias")
    g.n_params += pw.size + pb.size
    summed = g.buf(x.h, x.w, c)
    for h in range(nh):
        v = qkv.slice(h * (2 * kd + hd) + 2 * kd, hd)
        g.dwconv(v, 3, 1, f"{name}.attn.pe.conv.h{h}", act=ACT_NONE, out=summed.slice(h * hd, hd), res=att.slice(h * hd, hd),
                 weight=pw[h * hd:(h + 1) * hd], bias=pb[h * hd:(h + 1) * hd])
    b1 = g.conv(summed, c, 1, 1, f"{name}.attn.proj.conv", act=ACT_NONE, res=b, res_mode=RES_AFTER_ACT)          # b + attn(b)
    f = g.conv(b1, 2 * c, 1, 1, f"{name}.ffn.0.conv")
    # b + ffn(b) lands in b's own slot of the (a, b) buffer: the old b has no reader left, and cv2 reads cat(a, b) without a copy
    g.conv(f, c, 1, 1, f"{name}.ffn.1.conv", act=ACT_NONE, res=b1, res_mode=RES_AFTER_ACT, out=b)
    return g.conv(ab, x.c, 1, 1, f"{name}.cv2.conv", out=out)


def _cib(g, x, name, lk, out=None, shortcut=True):
    """CIB(c, c, e=1.0): dw3x3 -> 1x1 (2c) -> dw3x3 | fused RepVGGDW 7x7 -> 1x1 (c) -> dw3x3, all SiLU; x + cv1(x)."""
    c = x.c
    t = g.dwconv(x, 3, 1, f"{name}.cv1.0.conv")
    t = g.conv(t, 2 * c, 1, 1, f"{name}.cv1.1.conv")
    t = g.dwconv(t, 7, 1, f"{name}.cv1.2.conv.conv") if lk else g.dwconv(t, 3, 1, f"{name}.cv1.2.conv")
    t = g.conv(t, c, 1, 1, f"{name}.cv1.3.conv")
    return g.dwconv(t, 3, 1, f"{name}.cv1.4.conv", out=out, res=x if shortcut else None)


def _c2fcib(g, x, c2, n, shortcut, lk, name, out=None):
    c = c2 // 2
    cat = g.buf(x.h, x.w, (2 + n) * c)
    g.conv(x, 2 * c, 1, 1, f"{name}.cv1.conv", out=cat.slice(0, 2 * c))
    for i in range(n):
        _cib(g, cat.slice((1 + i) * c, c), f"{name}.m.{i}", lk, out=cat.slice((2 + i) * c, c), shortcut=shortcut)
    return g.conv(cat, c2, 1, 1, f"{name}.cv2.conv", out=out)


def yolov10(scale="n", nc=80, imgsz=640, wsrc=None, seed=0):
    depth, width, max_ch = V10_SCALES[scale]
    wsrc = wsrc or SynthWeights(seed, gain=synth_gain(f"yolov10{scale}"))
    H, W = _hw(imgsz)
    g = Graph(f"yolov10{scale}", 3, H, W, wsrc)
    ch = lambda c: _mk(c, width, max_ch)
    dep = lambda n: max(round(n * depth), 1)
    c1, c2, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    x, cin = g.input()
    cat12 = g.buf(H // 16, W // 16, c5 + c4)   # [up(10), 6]
    cat15 = g.buf(H // 8, W // 8, c4 + c3)     # [up(13), 4]
    cat18 = g.buf(H // 16, W // 16, c3 + c4)   # [17, 13]
    cat21 = g.buf(H // 32, W // 32, c4 + c5)   # [20, 10]
    x = g.conv(x, c1, 3, 2, "model.0.conv", true_cin=cin)
    x = g.conv(x, c2, 3, 2, "model.1.conv")
    x = _c2f(g, x, c2, dep(3), True, "model.2")
    x = g.conv(x, c3, 3, 2, "model.3.conv")
    p3b = _c2f(g, x, c3, dep(6), True, "model.4", out=cat15.slice(c4, c3))
    x = _scdown(g, p3b, c4, 3, 2, "model.5")
    p4b = _c2f(g, x, c4, dep(6), True, "model.6", out=cat12.slice(c5, c4))
    x = _scdown(g, p4b, c5, 3, 2, "model.7")
    x = _c2f(g, x, c5, dep(3), True, "model.8") if scale == "n" else _c2fcib(g, x, c5, dep(3), True, True, "model.8")
    x = _sppf(g, x, c5, "model.9")
    p5b = _psa_block(g, x, "model.10", out=cat21.slice(c4, c5))
    g.upsample2(p5b, out=cat12.slice(0, c5), name="model.11")
    n13 = _c2f(g, cat12, c4, dep(3), False, "model.13", out=cat18.slice(c3, c4))
    g.upsample2(n13, out=cat15.slice(0, c4), name="model.14")
    p3 = _c2f(g, cat15, c3, dep(3), False, "model.16")
    g.conv(p3, c3, 3, 2, "model.17.conv", out=cat18.slice(0, c3))
    p4 = _c2f(g, cat18, c4, dep(3), False, "model.19")
    _scdown(g, p4, c4, 3, 2, "model.20", out=cat21.slice(0, c4))
    p5 = _c2fcib(g, cat21, c5, dep(3), True, True, "model.22")
    # v10Detect, one-to-one branch (cv2: two 3x3 + 1x1 as v8; cv3: (dw3x3, 1x1), (dw3x3, 1x1), 1x1)
    feats = [p3, p4, p5]
    cb = max(16, feats[0].c // 4, 64)
    cc = max(feats[0].c, min(nc, 100))
    ins, strides = [], []
    for i, f in enumerate(feats):
        s = H // f.h
        strides.append(s)
        b = g.conv(f, cb, 3, 1, f"model.23.one2one_cv2.{i}.0.conv")
        b = g.conv(b, cb, 3, 1, f"model.23.one2one_cv2.{i}.1.conv")
        b = g.conv(b, 64, 1, 1, f"model.23.one2one_cv2.{i}.2", act=ACT_NONE, f32_out=True, bias_fill=1.0)
        c = g.dwconv(f, 3, 1, f"model.23.one2one_cv3.{i}.0.0.conv")
        c = g.conv(c, cc, 1, 1, f"model.23.one2one_cv3.{i}.0.1.conv")
        c = g.dwconv(c, 3, 1, f"model.23.one2one_cv3.{i}.1.0.conv")
        c = g.conv(c, cc, 1, 1, f"model.23.one2one_cv3.{i}.1.1.conv")
        c = g.conv(c, nc, 1, 1, f"model.23.one2one_cv3.{i}.2", act=ACT_NONE, f32_out=True,
                   bias_fill=math.log(5 / nc / (640 / s) ** 2))
        ins += [b, c]
    A = sum(f.h * f.w for f in feats)
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="model.23.decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    g.meta = dict(kind="yolov10", nc=nc, anchors=A, strides=strides)
    return g


# =====================================================================================
# YOLOv9t (GELAN-t: ultralytics yolov9t.yaml; WongKinYiu/yolov9 gelan-t), deploy form (RepConv re-parameterised to one 3x3 + SiLU).
# README.md:57 lists YOLOv9 among the detectors; yoloDetector.py:114,121 decodes its (1, 4+nc, A) head like v8's.  Modules:
# ELAN1, AConv (2x2 s1 average pool + 3x3 s2 conv), RepNCSPELAN4 (RepCSP = C3 of RepBottlenecks), SPPELAN, v8 Detect.
# =====================================================================================
def _repcsp(g, x, c2, n, name, out=None):
    """RepCSP(c1, c2, n): cv3(cat(m(cv1 x), cv2 x)), m = n RepBottlenecks (RepConv 3x3 -> Conv 3x3, + shortcut), e = 0.5."""
    c_ = c2 // 2
    T = g.buf(x.h, x.w, 3 * c_)                      # [m(cv1 x) | cv2 x | cv1 x], as in _c3
    cat = T.slice(0, 2 * c_)
    _, y = g.conv_siblings(x, [(c_, f"{name}.cv2.conv"), (c_, f"{name}.cv1.conv")], 1, 1, out=T.slice(c_, 2 * c_), draw=(1, 0))
    for i in range(n):
        t = g.conv(y, c_, 3, 1, f"{name}.m.{i}.cv1.conv")                       # RepConv in deploy form: one fused 3x3 + SiLU
        y = g.conv(t, c_, 3, 1, f"{name}.m.{i}.cv2.conv", out=cat.slice(0, c_) if i == n - 1 else None, res=y, res_mode=RES_AFTER_ACT)
    return g.conv(cat, c2, 1, 1, f"{name}.cv3.conv", out=out)


def _repncspelan4(g, x, c2, c3, c4, n, name, out=None):
    cat = g.buf(x.h, x.w, c3 + 2 * c4)
    g.conv(x, c3, 1, 1, f"{name}.cv1.conv", out=cat.slice(0, c3))
    t = _repcsp(g, cat.slice(c3 // 2, c3 // 2), c4, n, f"{name}.cv2.0")
    g.conv(t, c4, 3, 1, f"{name}.cv2.1.conv", out=cat.slice(c3, c4))
    t = _repcsp(g, cat.slice(c3, c4), c4, n, f"{name}.cv3.0")
    g.conv(t, c4, 3, 1, f"{name}.cv3.1.conv", out=cat.slice(c3 + c4, c4))
    return g.conv(cat, c2, 1, 1, f"{name}.cv4.conv", out=out)


def _elan1(g, x, c2, c3, c4, name, out=None):
    cat = g.buf(x.h, x.w, c3 + 2 * c4)
    g.conv(x, c3, 1, 1, f"{name}.cv1.conv", out=cat.slice(0, c3))
    g.conv(cat.slice(c3 // 2, c3 // 2), c4, 3, 1, f"{name}.cv2.conv", out=cat.slice(c3, c4))
    g.conv(cat.slice(c3, c4), c4, 3, 1, f"{name}.cv3.conv", out=cat.slice(c3 + c4, c4))
    return g.conv(cat, c2, 1, 1, f"{name}.cv4.conv", out=out)


def _aconv(g, x, c2, name, out=None):
    t = g.avgpool(x, 2, 1, 0, name=f"{name}.pool")
    return g.conv(t, c2, 3, 2, f"{name}.cv1.conv", out=out)


def _sppelan(g, x, c2, c3, name, out=None):
    cat = g.buf(x.h, x.w, 4 * c3)
    g.conv(x, c3, 1, 1, f"{name}.cv1.conv", out=cat.slice(0, c3))
    for i in range(3):
        g.maxpool(cat.slice(i * c3, c3), 5, 1, 2, out=cat.slice((i + 1) * c3, c3), name=f"{name}.m{i}")
    return g.conv(cat, c2, 1, 1, f"{name}.cv5.conv", out=out)


def _v9_detect(g, feats, nc, H, name):
    """The v8 Detect head of the YOLOv9 graphs (model.22)."""
    cb = max(16, feats[0].c // 4, 64)
    cc = max(feats[0].c, min(nc, 100))
    ins, strides = [], []
    for i, f in enumerate(feats):
        s = H // f.h
        strides.append(s)
        b = g.conv(f, cb, 3, 1, f"model.22.cv2.{i}.0.conv")
        b = g.conv(b, cb, 3, 1, f"model.22.cv2.{i}.1.conv")
        b = g.conv(b, 64, 1, 1, f"model.22.cv2.{i}.2", act=ACT_NONE, f32_out=True, bias_fill=1.0)
        c = g.conv(f, cc, 3, 1, f"model.22.cv3.{i}.0.conv")
        c = g.conv(c, cc, 3, 1, f"model.22.cv3.{i}.1.conv")
        c = g.conv(c, nc, 1, 1, f"model.22.cv3.{i}.2", act=ACT_NONE, f32_out=True, bias_fill=math.log(5 / nc / (640 / s) ** 2))
        ins += [b, c]
    A = sum(f.h * f.w for f in feats)
    head = g.buf(1, 1, (4 + nc) * A, f32=True)
    g._op(OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="model.22.decode")
    g.output(head, 0, [1, 4 + nc, A], "output0")
    g.meta = dict(kind="yolov9", nc=nc, anchors=A, strides=strides)
    return g


def yolov9t(nc=80, imgsz=640, wsrc=None, seed=0, scale="t"):
    """ultralytics cfg/models/v9/yolov9t.yaml / yolov9s.yaml: the s graph is the t graph with every width doubled (same modules, same
    repeats) -- `scale` "t" | "s"."""
    m = {"t": 1, "s": 2}[scale]
    name = "yolov9" + scale
    wsrc = wsrc or SynthWeights(seed, gain=synth_gain(name))
    H, W = _hw(imgsz)
    g = Graph(name, 3, H, W, wsrc)
    x, cin = g.input()
    cat11 = g.buf(H // 16, W // 16, m * (128 + 96))    # [up(9), 6]
    cat14 = g.buf(H // 8, W // 8, m * (96 + 64))       # [up(12), 4]
    cat17 = g.buf(H // 16, W // 16, m * (48 + 96))     # [16, 12]
    cat20 = g.buf(H // 32, W // 32, m * (64 + 128))    # [19, 9]
    x = g.conv(x, 16 * m, 3, 2, "model.0.conv", true_cin=cin)
    x = g.conv(x, 32 * m, 3, 2, "model.1.conv")
    x = _elan1(g, x, 32 * m, 32 * m, 16 * m, "model.2")
    x = _aconv(g, x, 64 * m, "model.3")
    p3b = _repncspelan4(g, x, 64 * m, 64 * m, 32 * m, 3, "model.4", out=cat14.slice(96 * m, 64 * m))
    x = _aconv(g, p3b, 96 * m, "model.5")
    p4b = _repncspelan4(g, x, 96 * m, 96 * m, 48 * m, 3, "model.6", out=cat11.slice(128 * m, 96 * m))
    x = _aconv(g, p4b, 128 * m, "model.7")
    x = _repncspelan4(g, x, 128 * m, 128 * m, 64 * m, 3, "model.8")
    p5b = _sppelan(g, x, 128 * m, 64 * m, "model.9", out=cat20.slice(64 * m, 128 * m))
    g.upsample2(p5b, out=cat11.slice(0, 128 * m), name="model.10")
    n12 = _repncspelan4(g, cat11, 96 * m, 96 * m, 48 * m, 3, "model.12", out=cat17.slice(48 * m, 96 * m))
    g.upsample2(n12, out=cat14.slice(0, 96 * m), name="model.13")
    p3 = _repncspelan4(g, cat14, 64 * m, 64 * m, 32 * m, 3, "model.15")
    _aconv(g, p3, 48 * m, "model.16", out=cat17.slice(0, 48 * m))
    p4 = _repncspelan4(g, cat17, 96 * m, 96 * m, 48 * m, 3, "model.18")
    _aconv(g, p4, 64 * m, "model.19", out=cat20.slice(0, 64 * m))
    p5 = _repncspelan4(g, cat20, 128 * m, 128 * m, 64 * m, 3, "model.21")
    return _v9_detect(g, [p3, p4, p5], nc, H, name)


def _adown(g, x, c2, name, out=None):
    """ADown(c1, c2): 2x2 s1 average pool, then channel halves -> (3x3 s2 conv | 3x3 s2 p1 max-pool + 1x1 conv), concatenated."""
    c = c2 // 2
    if out is None:
        out = g.buf(x.h // 2, x.w // 2, c2)
    t = g.avgpool(x, 2, 1, 0, name=f"{name}.pool")
    g.conv(t.slice(0, x.c // 2), c, 3, 2, f"{name}.cv1.conv", out=out.slice(0, c))
    mp = g.maxpool(t.slice(x.c // 2, x.c // 2), 3, 2, 1, name=f"{name}.mp")
    g.conv(mp, c, 1, 1, f"{name}.cv2.conv", out=out.slice(c, c))
    return out


def yolov9c(nc=80, imgsz=640, wsrc=None, seed=0):
    """ultralytics cfg/models/v9/yolov9c.yaml (GELAN-c, 25.3 M parameters / 102 GFLOPs fused): RepNCSPELAN4 with one RepBottleneck per
    RepCSP, ADown down-sampling, SPPELAN, v8 Detect."""
    wsrc = wsrc or SynthWeights(seed, gain=synth_gain("yolov9c"))
    H, W = _hw(imgsz)
    g = Graph("yolov9c", 3, H, W, wsrc)
    x, cin = g.input()
    cat11 = g.buf(H // 16, W // 16, 512 + 512)    # [up(9), 6]
    cat14 = g.buf(H // 8, W // 8, 512 + 512)      # [up(12), 4]
    cat17 = g.buf(H // 16, W // 16, 256 + 512)    # [16, 12]
    cat20 = g.buf(H // 32, W // 32, 512 + 512)    # [19, 9]
    x = g.conv(x, 64, 3, 2, "model.0.conv", true_cin=cin)
    x = g.conv(x, 128, 3, 2, "model.1.conv")
    x = _repncspelan4(g, x, 256, 128, 64, 1, "model.2")
    x = _adown(g, x, 256, "model.3")
    p3b = _repncspelan4(g, x, 512, 256, 128, 1, "model.4", out=cat14.slice(512, 512))
    x = _adown(g, p3b, 512, "model.5")
    p4b = _repncspelan4(g, x, 512, 512, 256, 1, "model.6", out=cat11.slice(512, 512))
    x = _adown(g, p4b, 512, "model.7")
    x = _repncspelan4(g, x, 512, 512, 256, 1, "model.8")
    p5b = _sppelan(g, x, 512, 256, "model.9", out=cat20.slice(512, 512))
    g.upsample2(p5b, out=cat11.slice(0, 512), name="model.10")
    n12 = _repncspelan4(g, cat11, 512, 512, 256, 1, "model.12", out=cat17.slice(256, 512))
    g.upsample2(n12, out=cat14.slice(0, 512), name="model.13")
    p3 = _repncspelan4(g, cat14, 256, 256, 128, 1, "model.15")
    _adown(g, p3, 256, "model.16", out=cat17.slice(0, 256))
    p4 = _repncspelan4(g, cat17, 512, 512, 256, 1, "model.18")
    _adown(g, p4, 512, "model.19", out=cat20.slice(0, 512))
    p5 = _repncspelan4(g, cat20, 512, 512, 256, 1, "model.21")
    return _v9_detect(g, [p3, p4, p5], nc, H, "yolov9c")


# =====================================================================================
# YOLOv6 v3.0 n / s (meituan/YOLOv6 configs/yolov6n.py, yolov6s.py; README.md:54 lists YOLOv6; yoloDetector.py:110-124 decodes its
# (1, A, 5+nc) head like v5's: probs = det[5:] * det[4], and EffiDeHead writes objectness 1).  Deploy form: every RepVGGBlock is one 3x3
# conv + ReLU.  EfficientRep backbone (stem, four ERBlocks = stride-2 RepVGGBlock + RepBlock, SimCSPSPPF), RepBiFPANNeck (BiFusion:
# ConvTranspose2d 2x2 up-sampling of the deeper map + 1x1 of the same-scale map + stride-2 3x3 of the shallower one, concatenated),
# EffiDeHead without DFL (1x1 stem, 3x3 cls / reg convs with SiLU, 1x1 predictors; anchor-free distances).  Weight names follow upstream's
# module paths; the deploy convs carry `.rbr_reparam` / `.block.conv`.
# Sizes: n 4.65 M parameters / 11.3 GFLOPs, s 18.54 M / 45.0 G (upstream tables: 4.7 M / 11.4 G, 18.5 M / 45.3 G).
# =====================================================================================
V6_SCALES = {"n": (0.33, 0.25), "s": (0.33, 0.50)}


def yolov6(scale="n", nc=80, imgsz=640, wsrc=None, seed=0):
    depth, width = V6_SCALES[scale]
    name = "yolov6" + scale
    wsrc = wsrc or SynthWeights(seed, gain=synth_gain(name))
    H, W = _hw(imgsz)
    g = Graph(name, 3, H, W, wsrc)
    x, cin = g.input()
    rep_n = lambda n: max(round(n * depth), 1) if n > 1 else n
    ch = [int(c * width) for c in (64, 128, 256, 512, 1024, 256, 128, 128, 256, 256, 512)]
    nb = [rep_n(n) for n in (1, 6, 12, 18, 6)]
    nn_ = rep_n(12)

    def rep(src, c, s, nm, out=None, true_cin=None):        # RepVGGBlock, deploy form
        return g.conv(src, c, 3, s, nm + ".rbr_reparam", act=ACT_RELU, out=out, true_cin=true_cin)

    def cbr(src, c, k, s, nm, out=None):                     # ConvBNReLU
        return g.conv(src, c, k, s, nm + ".block.conv", act=ACT_RELU, out=out)

    def repblock(src, c, n, nm, out=None):                   # RepBlock: conv1 + (n - 1) blocks
        y = rep(src, c, 1, nm + ".conv1", out=out if n == 1 else None)
        for i in range(n - 1):
            y = rep(y, c, 1, f"{nm}.block.{i}", out=out if i == n - 2 else None)
        return y

    # ---- EfficientRep
    x = rep(x, ch[0], 2, "backbone.stem", true_cin=cin)
    feats = []
    for i in range(1, 5):
        x = rep(x, ch[i], 2, f"backbone.ERBlock_{i + 1}.0")
        x = repblock(x, ch[i], nb[i], f"backbone.ERBlock_{i + 1}.1")
        feats.append(x)
    x3, x2, x1, x = feats                                     # P2 (stride 4), P3, P4, P5 before the SPP
    c_ = ch[4] // 2                                           # SimCSPSPPF(c5, c5, 5, e=0.5)
    sp = "backbone.ERBlock_5.2"
    cat2 = g.buf(x.h, x.w, 2 * c_)                            # cv7 input: cat(y0 = cv2(x), y3)
    cat4 = g.buf(x.h, x.w, 4 * c_)                            # cv5 input: cat(x1, m(x1), m(m(x1)), m(m(m(x1))))
    t = cbr(x, c_, 1, 1, sp + ".cv1")
    t = cbr(t, c_, 3, 1, sp + ".cv3")
    cbr(t, c_, 1, 1, sp + ".cv4", out=cat4.slice(0, c_))
    cbr(x, c_, 1, 1, sp + ".cv2", out=cat2.slice(0, c_))
    for i in range(3):
        g.maxpool(cat4.slice(i * c_, c_), 5, 1, 2, out=cat4.slice((i + 1) * c_, c_), name=f"{sp}.m{i}")
    t = cbr(cat4, c_, 1, 1, sp + ".cv5")
    cbr(t, c_, 3, 1, sp + ".cv6", out=cat2.slice(c_, c_))
    x0 = cbr(cat2, ch[4], 1, 1, sp + ".cv7")

    # ---- RepBiFPANNeck
    def bifusion(deep, same, shallow, c, nm):
        """BiFusion.forward: cv3(cat(upsample(x[0]), cv1(x[1]), downsample(cv2(x[2]))))."""
        cat = g.buf(same.h, same.w, 3 * c)
        g.deconv2x2(deep, c, nm + ".upsample.upsample_transpose", out=cat.slice(0, c))
        cbr(same, c, 1, 1, nm + ".cv1", out=cat.slice(c, c))
        t_ = cbr(shallow, c, 1, 1, nm + ".cv2")
        cbr(t_, c, 3, 2, nm + ".downsample", out=cat.slice(2 * c, c))
        return cbr(cat, c, 1, 1, nm + ".cv3")

    catn3 = g.buf(H // 16, W // 16, ch[7] + ch[6])            # cat(downsample2(pan_out2), fpn_out1)
    catn4 = g.buf(H // 32, W // 32, ch[9] + ch[5])            # cat(downsample1(pan_out1), fpn_out0)
    fpn0 = cbr(x0, ch[5], 1, 1, "neck.reduce_layer0", out=catn4.slice(ch[9], ch[5]))
    f0 = repblock(bifusion(fpn0, x1, x2, ch[5], "neck.Bifusion0"), ch[5], nn_, "neck.Rep_p4")
    fpn1 = cbr(f0, ch[6], 1, 1, "neck.reduce_layer1", out=catn3.slice(ch[7], ch[6]))
    pan2 = repblock(bifusion(fpn1, x2, x3, ch[6], "neck.Bifusion1"), ch[6], nn_, "neck.Rep_p3")
    cbr(pan2, ch[7], 3, 2, "neck.downsample2", out=catn3.slice(0, ch[7]))
    pan1 = repblock(catn3, ch[8], nn_, "neck.Rep_n3")
    cbr(pan1, ch[9], 3, 2, "neck.downsample1", out=catn4.slice(0, ch[9]))
    pan0 = repblock(catn4, ch[10], nn_, "neck.Rep_n4")

    # ---- EffiDeHead (inference, use_dfl = False)
    ins, strides = [], []
    levels = [pan2, pan1, pan0]
    for i, f in enumerate(levels):
        s_ = H // f.h
        strides.append(s_)
        st = g.conv(f, f.c, 1, 1, f"detect.stems.{i}.conv")                     # (layers in upstream's execution order: onnx_import
        cf = g.conv(st, f.c, 3, 1, f"detect.cls_convs.{i}.conv")               #  maps YOLOv6 weights by position)
        cl = g.conv(cf, nc, 1, 1, f"detect.cls_preds.{i}", act=ACT_NONE, f32_out=True, bias_fill=-math.log((1 - 0.01) / 0.01))
        rf = g.conv(st, f.c, 3, 1, f"detect.reg_convs.{i}.conv")
        rg = g.conv(rf, 4, 1, 1, f"detect.reg_preds.{i}", act=ACT_NONE, f32_out=True, bias_fill=1.0)
        ins += [rg, cl]
    A = sum(f.h * f.w for f in levels)
    no = nc + 5
    head = g.buf(1, 1, A * no, f32=True)
    g._op(OP_DETECT_V6, ins, head, params=[nc, A] + strides, name="detect.decode")
    g.output(head, 0, [1, A, no], "outputs")
    g.meta = dict(kind="yolov6", nc=nc, anchors=A, strides=strides)
    return g


# =====================================================================================
# YOLOv7-tiny (WongKinYiu/yolov7 cfg/deploy/yolov7-tiny.yaml; README.md:55 lists YOLOv7; yoloDetector.py:110-124 decodes its head as the
# v5 layout (1, A, 5+nc)).  78 rows, every Conv with LeakyReLU(0.1): ELAN-tiny blocks (two 1x1 branches, two chained 3x3, concat of the
# four, 1x1), MP = 2x2 stride-2 max-pool, an SPPCSPC-tiny (5 / 9 / 13 max-pools), PAN neck, IDetect (its ImplicitA / ImplicitM fold
# into the 1x1 at deploy: the v5 Detect arithmetic with the tiny yaml's anchors).  Weight names: `model.<row>.conv` as upstream.
# =====================================================================================
V7_TINY_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def yolov7_tiny(nc=80, imgsz=640, wsrc=None, seed=0):
    wsrc = wsrc or SynthWeights(seed, gain=synth_gain("yolov7-tiny"))
    H, W = _hw(imgsz)
    g = Graph("yolov7-tiny", 3, H, W, wsrc)
    x, cin = g.input()
    row = [0]

    def cv(src, c, k, s, out=None, true_cin=None):
        y = g.conv(src, c, k, s, f"model.{row[0]}.conv", act=ACT_LEAKY, out=out, true_cin=true_cin)
        row[0] += 1
        return y

    def elan(src, c, cout, out=None):
        """rows r .. r+5: 1x1 (a), 1x1 on the same input (b), 3x3 on b (c), 3x3 on c (d), Concat [d, c, b, a], 1x1."""
        cat = g.buf(src.h, src.w, 4 * c)
        b, _ = g.conv_siblings(src, [(c, f"model.{row[0] + 1}.conv"), (c, f"model.{row[0]}.conv")], 1, 1, out=cat.slice(2 * c, 2 * c), act=ACT_LEAKY, draw=(1, 0))
        row[0] += 2                                        # rows r (a -> slot 3) and r + 1 (b -> slot 2): one launch
        cc = cv(b, c, 3, 1, out=cat.slice(c, c))
        cv(cc, c, 3, 1, out=cat.slice(0, c))
        row[0] += 1                                        # the Concat row
        return cv(cat, cout, 1, 1, out=out)

    def mp(src):
        y = g.maxpool(src, 2, 2, 0, name=f"model.{row[0]}")
        row[0] += 1
        return y

    cat59 = g.buf(H // 16, W // 16, 256)                   # row 59: Concat [row 58, row 47]
    cat67 = g.buf(H // 32, W // 32, 512)                   # row 67: Concat [row 66, row 37]
    x = cv(x, 32, 3, 2, true_cin=cin)                      # 0
    x = cv(x, 64, 3, 2)                                    # 1
    x = elan(x, 32, 64)                                    # 2-7
    x = mp(x)                                              # 8
    p3b = elan(x, 64, 128)                                 # 9-14
    x = mp(p3b)                                            # 15
    p4b = elan(x, 128, 256)                                # 16-21
    x = mp(p4b)                                            # 22
    x = elan(x, 256, 512)                                  # 23-28
    # SPPCSPC-tiny, rows 29-37: bypass = row 29, pooled branch on row 30
    catb = g.buf(x.h, x.w, 512)                            # row 36: Concat [row 35, row 29]
    cats = g.buf(x.h, x.w, 1024)                           # row 34: Concat [SP13, SP9, SP5, row 30]
    cv(x, 256, 1, 1, out=catb.slice(256, 256))             # 29
    r30 = cv(x, 256, 1, 1, out=cats.slice(768, 256))       # 30
    for k, off in ((5, 512), (9, 256), (13, 0)):           # 31, 32, 33: SP(k) = MaxPool2d(k, 1, k // 2), each on row 30
        g.maxpool(r30, k, 1, k // 2, out=cats.slice(off, 256), name=f"model.{row[0]}")
        row[0] += 1
    row[0] += 1                                            # 34 Concat
    cv(cats, 256, 1, 1, out=catb.slice(0, 256))            # 35
    row[0] += 1                                            # 36 Concat
    p5 = cv(catb, 256, 1, 1, out=cat67.slice(256, 256))    # 37
    # top-down
    cat41 = g.buf(H // 16, W // 16, 256)                   # row 41: Concat [row 40, row 39]
    t = cv(p5, 128, 1, 1)                                  # 38
    g.upsample2(t, out=cat41.slice(128, 128), name=f"model.{row[0]}"); row[0] += 1      # 39
    cv(p4b, 128, 1, 1, out=cat41.slice(0, 128))            # 40 (route backbone P4 = row 21)
    row[0] += 1                                            # 41
    n47 = elan(cat41, 64, 128, out=cat59.slice(128, 128))  # 42-47
    cat51 = g.buf(H // 8, W // 8, 128)                     # row 51: Concat [row 50, row 49]
    t = cv(n47, 64, 1, 1)                                  # 48
    g.upsample2(t, out=cat51.slice(64, 64), name=f"model.{row[0]}"); row[0] += 1        # 49
    cv(p3b, 64, 1, 1, out=cat51.slice(0, 64))              # 50 (route backbone P3 = row 14)
    row[0] += 1                                            # 51
    n57 = elan(cat51, 32, 64)                              # 52-57
    # bottom-up
    cv(n57, 128, 3, 2, out=cat59.slice(0, 128))            # 58
    row[0] += 1                                            # 59
    n65 = elan(cat59, 64, 128)                             # 60-65
    cv(n65, 256, 3, 2, out=cat67.slice(0, 256))            # 66
    row[0] += 1                                            # 67
    n73 = elan(cat67, 128, 256)                            # 68-73
    assert row[0] == 74, row[0]
    feats = [cv(n57, 128, 3, 1), cv(n65, 256, 3, 1), cv(n73, 512, 3, 1)]      # 74, 75, 76
    no = nc + 5
    ins, strides = [], []
    for i, f in enumerate(feats):                          # 77: IDetect, deploy form (implicit layers folded into the 1x1)
        s_ = H // f.h
        strides.append(s_)
        bias = np.zeros((3, no), np.float32)
        bias[:, 4] = math.log(8 / (640 / s_) ** 2)
        bias[:, 5:] = math.log(0.6 / (nc - 0.999999))
        name = f"model.77.m.{i}"
        if isinstance(wsrc, SynthWeights) and name + ".bias" not in wsrc.store:
            wsrc.store[name + ".bias"] = bias.reshape(-1) + 0.01 * wsrc.rng.standard_normal(3 * no).astype(np.float32)
        ins.append(g.conv(f, 3 * no, 1, 1, name, act=ACT_NONE, f32_out=True))
    A = 3 * sum(f.h * f.w for f in feats)
    head = g.buf(1, 1, A * no, f32=True)
    g._op(OP_DETECT_V5, ins, head, params=[nc, A] + strides, name="model.77.decode")
    g.ops[-1]["w"] = g._blob(np.asarray([a for lvl in V7_TINY_ANCHORS for a in lvl], np.float32))
    g.output(head, 0, [1, A, no], "output0")
    g.meta = dict(kind="yolov7", nc=nc, anchors=A, strides=strides)
    return g


# =====================================================================================
# YOLOv5 v6.2
# =====================================================================================
V5_SCALES = {"n": (0.33, 0.25), "s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}
V5_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]  # yoloDetector.py:23


def _c3(g, x, c2, n, shortcut, name, out=None):
    c_ = c2 // 2
    T = g.buf(x.h, x.w, 3 * c_)                      # [m(cv1 x) | cv2 x | cv1 x]: cv3 reads the first two slots, cv1 + cv2 are one launch
    cat = T.slice(0, 2 * c_)
    _, y = g.conv_siblings(x, [(c_, f"{name}.cv2.conv"), (c_, f"{name}.cv1.conv")], 1, 1, out=T.slice(c_, 2 * c_), draw=(1, 0))
    for i in range(n):
        t = g.conv(y, c_, 1, 1, f"{name}.m.{i}.cv1.conv")
        last = (i == n - 1)
        y = g.conv(t, c_, 3, 1, f"{name}.m.{i}.cv2.conv", out=cat.slice(0, c_) if last else None,
                   res=y if shortcut else None, res_mode=RES_AFTER_ACT if shortcut else RES_NONE)
    return g.conv(cat, c2, 1, 1, f"{name}.cv3.conv", out=out)


def yolov5(scale="n", nc=80, imgsz=640, wsrc=None, seed=0):
    depth, width = V5_SCALES[scale]
    wsrc = wsrc or SynthWeights(seed, gain=V5_SILU_GAIN)
    H, W = _hw(imgsz)
    g = Graph(f"yolov5{scale}", 3, H, W, wsrc)
    ch = lambda c: int(math.ceil(c * width / 8) * 8)
    dep = lambda n: max(round(n * depth), 1)
    c1, c2, c3, c4, c5 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    x, cin = g.input()
    cat12 = g.buf(H // 16, W // 16, c4 + c4)   # [up(10), 6]
    cat16 = g.buf(H // 8, W // 8, c3 + c3)     # [up(14), 4]
    cat19 = g.buf(H // 16, W // 16, c3 + c3)   # [18, 14]
    cat22 = g.buf(H // 32, W // 32, c4 + c4)   # [21, 10]
    x = g.conv(x, c1, 6, 2, "model.0.conv", true_cin=cin, pad=2)
    x = g.conv(x, c2, 3, 2, "model.1.conv")
    x = _c3(g, x, c2, dep(3), True, "model.2")
    x = g.conv(x, c3, 3, 2, "model.3.conv")
    b4 = _c3(g, x, c3, dep(6), True, "model.4", out=cat16.slice(c3, c3))
    x = g.conv(b4, c4, 3, 2, "model.5.conv")
    b6 = _c3(g, x, c4, dep(9), True, "model.6", out=cat12.slice(c4, c4))
    x = g.conv(b6, c5, 3, 2, "model.7.conv")
    x = _c3(g, x, c5, dep(3), True, "model.8")
    x = _sppf(g, x, c5, "model.9")
    n10 = g.conv(x, c4, 1, 1, "model.10.conv", out=cat22.slice(c4, c4))
    g.upsample2(n10, out=cat12.slice(0, c4), name="model.11")
    x = _c3(g, cat12, c4, dep(3), False, "model.13")
    n14 = g.conv(x, c3, 1, 1, "model.14.conv", out=cat19.slice(c3, c3))
    g.upsample2(n14, out=cat16.slice(0, c3), name="model.15")
    p3 = _c3(g, cat16, c3, dep(3), False, "model.17")
    g.conv(p3, c3, 3, 2, "model.18.conv", out=cat19.slice(0, c3))
    p4 = _c3(g, cat19, c4, dep(3), False, "model.20")
    g.conv(p4, c4, 3, 2, "model.21.conv", out=cat22.slice(0, c4))
    p5 = _c3(g, cat22, c5, dep(3), False, "model.23")
    feats = [p3, p4, p5]
    no = nc + 5
    ins, strides = [], []
    for i, f in enumerate(feats):
        s = H // f.h
        strides.append(s)
        # upstream bias init: obj log(8/(640/s)^2), cls log(0.6/(nc-0.999999)); synthetic -> keep scores sparse
        bias = np.zeros((3, no), np.float32)
        bias[:, 4] = math.log(8 / (640 / s) ** 2)
        bias[:, 5:] = math.log(0.6 / (nc - 0.999999))
        name = f"model.24.m.{i}"
        if isinstance(wsrc, SynthWeights) and name + ".bias" not in wsrc.store:
            wsrc.store[name + ".bias"] = bias.reshape(-1) + 0.01 * wsrc.rng.standard_normal(3 * no).astype(np.float32)
        ins.append(g.conv(f, 3 * no, 1, 1, name, act=ACT_NONE, f32_out=True))
    A = 3 * sum(f.h * f.w for f in feats)
    head = g.buf(1, 1, A * no, f32=True)
    anchors = [a for lvl in V5_ANCHORS for a in lvl]
    g._op(OP_DETECT_V5, ins, head, params=[nc, A] + strides, name="model.24.decode")
    g.ops[-1]["w"] = g._blob(np.asarray(anchors, np.float32))
    g.output(head, 0, [1, A, no], "output0")
    g.meta = dict(kind="yolov5", nc=nc, anchors=A, strides=strides)
    return g


# =====================================================================================
# UFLDv2 (CULane): torchvision ResNet-18/34 topology + parsingNet head
# =====================================================================================
RESNET_DEPTHS = {"18": [2, 2, 2, 2], "34": [3, 4, 6, 3]}


def ufldv2(backbone="18", in_h=320, in_w=1600, num_grid_row=200, num_cls_row=72, num_grid_col=100, num_cls_col=81,
           num_lanes=4, fc_norm=True, wsrc=None, seed=0):
    wsrc = wsrc or SynthWeights(seed, gain=RELU_RES_GAIN)
    g = Graph(f"ufldv2_culane_res{backbone}", 3, in_h, in_w, wsrc)
    x, cin = g.input()
    x = g.conv(x, 64, 7, 2, "model.conv1", act=ACT_RELU, true_cin=cin, pad=3)      # conv1+bn1+relu (backbone.py:50-52)
    x = g.maxpool(x, 3, 2, 1, name="model.maxpool")                                # :53
    cinp = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], RESNET_DEPTHS[backbone])):
        for bi in range(nblk):
            s = 2 if (li > 0 and bi == 0) else 1
            name = f"model.layer{li + 1}.{bi}"
            idt = x
            if s != 1 or cinp != planes:
                idt = g.conv(x, planes, 1, s, f"{name}.downsample.0", act=ACT_NONE, pad=0)
            t = g.conv(x, planes, 3, s, f"{name}.conv1", act=ACT_RELU)
            x = g.conv(t, planes, 3, 1, f"{name}.conv2", act=ACT_RELU, res=idt, res_mode=RES_BEFORE_ACT)
            cinp = planes
    fea = g.conv(x, 8, 1, 1, "pool", act=ACT_NONE, f32_out=fc_norm, pad=0)        # model_culane.py:39,48
    input_dim = in_h // 32 * in_w // 32 * 8                                        # :23
    assert fea.h * fea.w * 8 == input_dim
    mid = 2048
    dims = [num_grid_row * num_cls_row * num_lanes, num_grid_col * num_cls_col * num_lanes,
            2 * num_cls_row * num_lanes, 2 * num_cls_col * num_lanes]
    total = sum(dims)
    # torch flattens (C,H,W); our activation is (H,W,C): permute LN affine + FC1 input columns once, offline
    hw = fea.h * fea.w
    perm = (np.arange(8)[None, :] * hw + np.arange(hw)[:, None]).reshape(-1)      # ours[j] = torch[perm[j]]
    x = fea                                                                        # LayerNorm reads it flat (h*w*c)
    if not fc_norm:                                                                # Tusimple: cls.0 = Identity (configs/tusimple_res18.py:35)
        x = g.alias(fea, 1, 1, input_dim)
    if fc_norm:                                                                    # cls.0 LayerNorm (:34)
        lw = wsrc("cls.0.weight", (input_dim,), "ln_w")
        lb = wsrc("cls.0.bias", (input_dim,), "ln_b")
        ln = g.buf(1, 1, input_dim)
        woff = g._blob(lw[perm]); boff = g._blob(lb[perm])
        g._op(OP_LAYERNORM, [x], ln, w=woff, b=boff, params=[1e-5], name="cls.0")
        g.n_params += 2 * input_dim
        x = ln
    W1 = wsrc("cls.1.weight", (mid, input_dim), "linear")
    perm_src = DictWeights({"cls.1.weight.perm": np.ascontiguousarray(W1[:, perm])})
    keep, g.w = g.w, perm_src
    h1 = g.conv(x, mid, 1, 1, "cls.1", act=ACT_RELU, wname="cls.1.weight.perm", bias=False, wkind="linear")
    g.w = keep
    b1 = wsrc("cls.1.bias", (mid,), "bias")
    g.ops[-1]["b"] = g._blob(b1)
    g.n_params += mid
    out = g.conv(h1, total, 1, 1, "cls.3", act=ACT_NONE, f32_out=True, wkind="linear")
    off = 0
    shapes = [[1, num_grid_row, num_cls_row, num_lanes], [1, num_grid_col, num_cls_col, num_lanes],
              [1, 2, num_cls_row, num_lanes], [1, 2, num_cls_col, num_lanes]]
    for nm, d, shp in zip(("loc_row", "loc_col", "exist_row", "exist_col"), dims, shapes):   # :56-59
        g.output(out, off, shp, nm)
        off += d
    g.meta = dict(kind="ufldv2", dims=dims, total=total, fc_norm=fc_norm)
    return g


def ufld_v1(backbone="18", in_h=288, in_w=800, griding_num=100, cls_num_per_lane=56, num_lanes=4, wsrc=None, seed=0):
    """UFLD (v1) parsingNet: ResNet trunk -> 1x1 `pool` conv 512->8 -> view(-1, 1800) -> Linear 2048 -> ReLU ->
    Linear (G+1)*K*L -> one (1, G+1, K, L) tensor.  Follows the network source the reference vendors for its ONNX export
    (TrafficLaneDetector/ufldDetector/exportLib/ultrafastLane/model.py:19-89, backbone.py); the detector side pins the single
    output and its layout (ultrafastLaneDetector.py:73-75,96-109) and the 800x288 input (:84)."""
    wsrc = wsrc or SynthWeights(seed, gain=RELU_RES_GAIN)
    g = Graph(f"ufld_v1_res{backbone}", 3, in_h, in_w, wsrc)
    x, cin = g.input()
    x = g.conv(x, 64, 7, 2, "model.conv1", act=ACT_RELU, true_cin=cin, pad=3)
    x = g.maxpool(x, 3, 2, 1, name="model.maxpool")
    cinp = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], RESNET_DEPTHS[backbone])):
        for bi in range(nblk):
            s = 2 if (li > 0 and bi == 0) else 1
            name = f"model.layer{li + 1}.{bi}"
            idt = x
            if s != 1 or cinp != planes:
                idt = g.conv(x, planes, 1, s, f"{name}.downsample.0", act=ACT_NONE, pad=0)
            t = g.conv(x, planes, 3, s, f"{name}.conv1", act=ACT_RELU)
            x = g.conv(t, planes, 3, 1, f"{name}.conv2", act=ACT_RELU, res=idt, res_mode=RES_BEFORE_ACT)
            cinp = planes
    fea = g.conv(x, 8, 1, 1, "pool", act=ACT_NONE, pad=0)
    input_dim = fea.h * fea.w * 8                                                   # 1800 at 288x800
    mid, total = 2048, (griding_num + 1) * cls_num_per_lane * num_lanes
    hw = fea.h * fea.w
    perm = (np.arange(8)[None, :] * hw + np.arange(hw)[:, None]).reshape(-1)       # ours (h,w,c) <- torch (c,h,w)
    x = g.alias(fea, 1, 1, input_dim)
    W1 = wsrc("cls.0.weight", (mid, input_dim), "linear")
    keep, g.w = g.w, DictWeights({"cls.0.weight.perm": np.ascontiguousarray(W1[:, perm])})
    h1 = g.conv(x, mid, 1, 1, "cls.0", act=ACT_RELU, wname="cls.0.weight.perm", bias=False, wkind="linear")
    g.w = keep
    g.ops[-1]["b"] = g._blob(wsrc("cls.0.bias", (mid,), "bias"))
    g.n_params += mid
    out = g.conv(h1, total, 1, 1, "cls.2", act=ACT_NONE, f32_out=True, wkind="linear")
    g.output(out, 0, [1, griding_num + 1, cls_num_per_lane, num_lanes], "output")
    g.meta = dict(kind="ufld_v1", total=total)
    return g


UFLD1_CULANE = dict(griding_num=200, cls_num_per_lane=18)
TUSIMPLE = dict(in_h=320, in_w=800, num_grid_row=100, num_cls_row=56, num_grid_col=100, num_cls_col=41, fc_norm=False)
# CurveLanes configuration (configs/curvelanes_res18.py:25-36): 1600x800 input, 200/100 grid cells, 72/41 anchors, 10 lanes, LayerNorm.
# The reference exports every UFLDv2 config through model_culane.parsingNet (convertPytorchToONNX.py:65-70: the model_curvelanes
# branch is commented out "TODO : not done"), so this is the same graph at another geometry: 25x50 layer4 maps, Linear 10000 -> 2048
# -> 187,260.
CURVELANES = dict(in_h=800, in_w=1600, num_grid_row=200, num_cls_row=72, num_grid_col=100, num_cls_col=41, num_lanes=10, fc_norm=True)

# =====================================================================================
# EfficientDet-D0  (Tan, Pang, Le: "EfficientDet", arXiv:1911.09070; the reference's EfficientdetDetector loads an exported
# efficientdet-d0 graph: ObjectDetector/efficientdetDetector.py:18-44, demo default 'models/efficientdet-d0-coco_fp32.onnx' :119)
# =====================================================================================
# EfficientNet-B0 stages (Tan & Le, arXiv:1905.11946 table 1): (expand ratio, kernel, stride, out channels, repeats); SE ratio 0.25
# of the block's INPUT channels; swish everywhere; identity skip when stride 1 and in == out.
EFFNET_B0 = [(1, 3, 1, 16, 1), (6, 3, 2, 24, 2), (6, 5, 2, 40, 2), (6, 3, 2, 80, 3), (6, 5, 1, 112, 3), (6, 5, 2, 192, 4), (6, 3, 1, 320, 1)]
EFFDET_D0 = dict(imgsz=512, fpn_c=64, fpn_cells=3, head_layers=3, num_anchors=9)
EFFDET_GAIN = 1.0     # synthetic weights: critical gain ~1.02 on camera-like 512 x 512 inputs (class logits reach +-30 there, box regressions
                      # exp() to infinity; 1.1 explodes through the linear project / BiFPN convs).  At 1.0 the per-anchor best logit spreads
                      # over 0.5 (median) .. 2 (99 %) .. 4.6 (max) around the header bias and regressions stay |d| < 3 (a trained checkpoint
                      # needs none of this)


def fusion_weights(p):
    """BiFPN fast normalised fusion (EfficientDet eq. 3) at inference: relu(p_i) / (sum_j relu(p_j) + 1e-4), float32."""
    w = np.maximum(np.asarray(p, np.float32), np.float32(0))
    return (w / (w.sum(dtype=np.float32) + np.float32(1e-4))).astype(np.float32)


def _mbconv(g, x, name, expand, k, s, cout):
    cin = x.c
    t = x
    if expand != 1:
        t = g.conv(t, cin * expand, 1, 1, name + ".expand")
    t = g.dwconv(t, k, s, name + ".dw")
    t = g.se(t, max(1, int(cin * 0.25)), name + ".se")
    skip = x if (s == 1 and cin == cout) else None
    return g.conv(t, cout, 1, 1, name + ".project", act=ACT_NONE, res=skip, res_mode=RES_AFTER_ACT if skip is not None else RES_NONE)


def _sepconv(g, x, cout, name, act=ACT_NONE, f32_out=False, dw_name=None, bias_fill=None):
    """SeparableConvBlock: depth-wise 3x3 (no bias) -> point-wise 1x1 (bias; the BatchNorm behind it folded in) [-> swish]."""
    c = x.c
    dw = g.w((dw_name or name) + ".dw.weight", (c, 1, 3, 3), "conv")
    t = g.dwconv(x, 3, 1, name + ".dw", act=ACT_NONE, weight=dw, bias=np.zeros(c, np.float32))
    return g.conv(t, cout, 1, 1, name + ".pw", act=act, f32_out=f32_out, bias_fill=bias_fill)


def efficientdet(nc=90, imgsz=512, wsrc=None, seed=0, fpn_c=64, fpn_cells=3, head_layers=3, num_anchors=9, cls_bias=-5.0):
    """EfficientDet-D0 up to its two raw head tensors per pyramid level: box regression (dy, dx, dh, dw) x 9 anchors and class logits
    nc x 9 anchors, rows ordered (y, x, anchor) -- what the exported graph feeds its in-graph anchor decode + NMS
    (postproc.EffdetTail).  Symmetric k // 2 padding (the PyTorch-native variant of the architecture): weights of the public
    TF-'same'-padded checkpoints (stride 2 pads right / bottom only) are NOT valid for this graph -- see detectors.EfficientdetDetector.
    cls_bias: the classifier
    header's bias (trained nets start it at -log(99) = -4.6; -5 leaves ~1 % of the seeded net's anchors over a 0.05 score threshold)."""
    H, W = _hw(imgsz)
    assert H % 128 == 0 and W % 128 == 0, "EfficientDet needs inputs divisible by 128 (five pyramid levels, 2x resampling)"
    ws = wsrc or SynthWeights(seed, gain=EFFDET_GAIN)
    g = Graph("efficientdet-d0", 3, H, W, ws)
    x, c3 = g.input()
    t = g.conv(x, 32, 3, 2, "stem", true_cin=c3)
    feats, bi = [], 0
    for si, (e, k, s, c, n) in enumerate(EFFNET_B0):
        for r in range(n):
            t = _mbconv(g, t, f"blocks.{bi}", e, k, s if r == 0 else 1, c)
            bi += 1
        if si in (2, 4, 6):
            feats.append(t)
    c3f, c4f, c5f = feats
    # ---- BiFPN
    p = None
    for cell in range(fpn_cells):
        nm = f"bifpn.{cell}"
        if cell == 0:
            p3 = g.conv(c3f, fpn_c, 1, 1, nm + ".p3_down", act=ACT_NONE)
            p4 = g.conv(c4f, fpn_c, 1, 1, nm + ".p4_down", act=ACT_NONE)
            p5 = g.conv(c5f, fpn_c, 1, 1, nm + ".p5_down", act=ACT_NONE)
            p6 = g.maxpool(g.conv(c5f, fpn_c, 1, 1, nm + ".p5_to_p6", act=ACT_NONE), 3, 2, 1, name=nm + ".p6_pool")
            p7 = g.maxpool(p6, 3, 2, 1, name=nm + ".p7_pool")
            p4b = g.conv(c4f, fpn_c, 1, 1, nm + ".p4_down_2", act=ACT_NONE)   # the bottom-up path of the first cell re-projects P4 / P5
            p5b = g.conv(c5f, fpn_c, 1, 1, nm + ".p5_down_2", act=ACT_NONE)
        else:
            p3, p4, p5, p6, p7 = p
            p4b, p5b = p4, p5

        def fw(tag, n):
            return fusion_weights(ws(f"{nm}.{tag}", (n,), "ln_w"))

        p6u = _sepconv(g, g.wsum([p6, p7], fw("p6_w1", 2), nm + ".p6_td"), fpn_c, nm + ".conv6_up")
        p5u = _sepconv(g, g.wsum([p5, p6u], fw("p5_w1", 2), nm + ".p5_td"), fpn_c, nm + ".conv5_up")
        p4u = _sepconv(g, g.wsum([p4, p5u], fw("p4_w1", 2), nm + ".p4_td"), fpn_c, nm + ".conv4_up")
        p3o = _sepconv(g, g.wsum([p3, p4u], fw("p3_w1", 2), nm + ".p3_td"), fpn_c, nm + ".conv3_up")
        p4o = _sepconv(g, g.wsum([p4b, p4u, g.maxpool(p3o, 3, 2, 1, name=nm + ".p3_ds")], fw("p4_w2", 3), nm + ".p4_bu"), fpn_c, nm + ".conv4_down")
        p5o = _sepconv(g, g.wsum([p5b, p5u, g.maxpool(p4o, 3, 2, 1, name=nm + ".p4_ds")], fw("p5_w2", 3), nm + ".p5_bu"), fpn_c, nm + ".conv5_down")
        p6o = _sepconv(g, g.wsum([p6, p6u, g.maxpool(p5o, 3, 2, 1, name=nm + ".p5_ds")], fw("p6_w2", 3), nm + ".p6_bu"), fpn_c, nm + ".conv6_down")
        p7o = _sepconv(g, g.wsum([p7, g.maxpool(p6o, 3, 2, 1, name=nm + ".p6_ds")], fw("p7_w2", 2), nm + ".p7_bu"), fpn_c, nm + ".conv7_down")
        p = (p3o, p4o, p5o, p6o, p7o)
    # ---- heads: the depth-wise / point-wise convs of a layer are shared by the five levels, the BatchNorm behind them is not (folded:
    # point-wise parameters per level); the header has no BatchNorm
    for lv, f in enumerate(p):
        for branch, cout in (("regressor", num_anchors * 4), ("classifier", num_anchors * nc)):
            t = f
            for i in range(head_layers):
                t = _sepconv(g, t, fpn_c, f"{branch}.l{lv}.{i}", act=ACT_SILU, dw_name=f"{branch}.conv_list.{i}")
            o = _sepconv(g, t, cout, f"{branch}.l{lv}.header", f32_out=True, dw_name=f"{branch}.header",
                         bias_fill=cls_bias if branch == "classifier" else None)
            per = 4 if branch == "regressor" else nc
            g.output(o, 0, [1, o.h * o.w * num_anchors, per], ("regression" if branch == "regressor" else "classification") + f".l{lv}")
    return g


BUILDERS = {
    "efficientdet-d0": lambda **k: efficientdet(**k),
    "yolov8n": lambda **k: yolov8("n", **k), "yolov8s": lambda **k: yolov8("s", **k),
    "yolov8m": lambda **k: yolov8("m", **k), "yolov8l": lambda **k: yolov8("l", **k),
    "yolov8x": lambda **k: yolov8("x", **k),
    "yolov10n": lambda **k: yolov10("n", **k),
    "yolov10s": lambda **k: yolov10("s", **k),
    "yolov9t": lambda **k: yolov9t(**k),
    "yolov9s": lambda **k: yolov9t(scale="s", **k),
    "yolov9c": lambda **k: yolov9c(**k),
    "yolov7-tiny": lambda **k: yolov7_tiny(**k),
    "yolov6n": lambda **k: yolov6("n", **k),
    "yolov6s": lambda **k: yolov6("s", **k),
    "yolov5n": lambda **k: yolov5("n", **k), "yolov5s": lambda **k: yolov5("s", **k),
    "yolov5m": lambda **k: yolov5("m", **k), "yolov5l": lambda **k: yolov5("l", **k), "yolov5x": lambda **k: yolov5("x", **k),
    "ufldv2_res18": lambda **k: ufldv2("18", **k), "ufldv2_res34": lambda **k: ufldv2("34", **k),
    # Tusimple configuration (configs/tusimple_res18.py:28-35): 800x320 input, 100/100 grid cells, 56/41 anchors, no LayerNorm
    # UFLD v1 (ultrafastLaneDetector.py): Tusimple 100 cells x 56 anchors, CULane 200 x 18, both 800x288
    "ufld_v1_res18": lambda **k: ufld_v1("18", **k), "ufld_v1_res34": lambda **k: ufld_v1("34", **k),
    "ufld_v1_culane_res18": lambda **k: ufld_v1("18", **dict(UFLD1_CULANE, **k)),
    "ufldv2_tusimple_res18": lambda **k: ufldv2("18", **dict(TUSIMPLE, **k)), "ufldv2_tusimple_res34": lambda **k: ufldv2("34", **dict(TUSIMPLE, **k)),
    "ufldv2_curvelanes_res18": lambda **k: ufldv2("18", **dict(CURVELANES, **k)), "ufldv2_curvelanes_res34": lambda **k: ufldv2("34", **dict(CURVELANES, **k)),
}



def build(name, **kw):
    return BUILDERS[name](**kw)
