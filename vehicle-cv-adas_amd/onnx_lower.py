"""ONNX graph -> engine op list: load detector graphs `models.py` does not hand-build (coreEngine.py:159-186 hands ANY
ONNX file to ONNXRuntime).

`onnx_import.detect_arch` recognises the architectures with a hand-written builder and pours the file's weights into them.  Everything
else -- another width or depth of a supported family (YOLOv8 at a custom scale, YOLOv7 / v5-layout variants, GELAN-style stacks), an
ad-hoc CSP network -- goes through this module: the node list is walked once and mapped onto the op list the engine already executes,

    Conv [+ Sigmoid, Mul | Relu | LeakyRelu(0.1)] [+ Add]      -> OP_CONV with a fused activation and residual (either order)
    Conv with group == channels                                 -> OP_DWCONV
    HardSwish | x * HardSigmoid(1/6) | HardSigmoid(1/6) | Clip(0, 6)   -> a one-input OP_WSUM (a stand-alone activation layer: the conv in front keeps ACT_NONE)
    GlobalAveragePool -> Conv -> swish | Relu -> Conv -> Sigmoid | HardSigmoid -> Mul(x, .)   -> OP_SE_GATE + OP_SCALE
    MaxPool / AveragePool(count_include_pad) / Resize(nearest x2) / ConvTranspose(k 2, s 2)
    Concat(axis 1) / Split(axis 1) / Slice(axis 1)              -> no op at all: producers write channel slices of one buffer, consumers read views
    the Detect tail                                             -> OP_DETECT_V8 (three Concat[box, cls] -> Reshape pairs feeding one axis-2 Concat;
                                                                   everything behind it -- DFL, dist2bbox, sigmoid -- is the op) or
                                                                   OP_DETECT_V5 (three 1x1 convs of 3 (5 + nc) channels reshaped to (1, 3, no, h, w);
                                                                   anchors read from the graph's anchor_grid constants)

and the engine's own load-time passes (stem, pair / C2f / Detect fusion, folded upsamples, conv_halo8 / conv_h8x3 selection) then see
the same op list a hand-written builder would have produced.  Anything outside this vocabulary raises ValueError naming the node --
softmax attention (YOLOv10's PSA) and stand-alone element-wise arithmetic among them; those graphs need a builder.

Buffer planning: a Concat's output is ONE buffer and each input tensor is assigned the channel slice it occupies, so its producer writes
there directly (a tensor that is already placed elsewhere, or a graph input, is copied in by a 1x1 max-pool).  Split / Slice outputs are
views of their source; consecutive slices of one tensor that re-appear in order inside a Concat (C2f: cv1's two halves) place the whole
source tensor.  Concats may nest."""
import numpy as np

try:
    from . import models as M
except ImportError:  # executed top-level
    import models as M


class LowerError(ValueError):
    pass


def _const(m, name):
    return m.initializers.get(name)


def _ints(m, nd, attr, inp):
    """An integer list given as attribute `attr` (older opsets) or as constant input number `inp` (newer)."""
    if attr in nd["attrs"] and nd["attrs"][attr] is not None:
        v = nd["attrs"][attr]
        return [int(x) for x in (v if isinstance(v, (list, tuple)) else [v])]
    if inp is not None and len(nd["inputs"]) > inp and nd["inputs"][inp]:
        c = _const(m, nd["inputs"][inp])
        if c is None:
            raise LowerError("node %s (%s): input %d must be a constant" % (nd["name"], nd["op"], inp))
        return [int(x) for x in np.asarray(c).reshape(-1)]
    return None


class _Lowering:
    def __init__(self, m, name):
        self.m = m
        self.nodes = m.nodes
        self.name = name
        if len(m.inputs) != 1 or len(m.inputs[0][1]) != 4:
            raise LowerError("expected one NCHW graph input, found %s" % (m.inputs,))
        self.in_name, ishape = m.inputs[0]
        if any((not isinstance(d, int)) or d <= 0 for d in ishape[1:]):
            raise LowerError("dynamic input size %s (export with fixed H and W)" % (ishape,))
        self.in_c, self.in_h, self.in_w = ishape[1], ishape[2], ishape[3]
        if self.in_c > 8:
            raise LowerError("input has %d channels (the engine's input layer takes up to 8)" % self.in_c)
        self.shape = {self.in_name: (self.in_c, self.in_h, self.in_w)}      # activation tensors: (C, H, W)
        self.producer = {}
        self.consumers = {}
        for i, nd in enumerate(self.nodes):
            for o in nd["outputs"]:
                self.producer[o] = i
            for t in nd["inputs"]:
                if t and t not in m.initializers:
                    self.consumers.setdefault(t, []).append(i)
        self.graph_outs = [n for n, _ in m.outputs]

    # ------------------------------------------------------------------ pass 1a: macro ops
    def _single_consumer(self, t, op=None):
        c = self.consumers.get(t, [])
        if len(c) != 1 or t in self.graph_outs:
            return None
        nd = self.nodes[c[0]]
        return c[0] if (op is None or nd["op"] == op) else None

    def _absorb_act(self, t, used):
        """(act, output tensor) of the activation nodes that consume conv output t, marking them used."""
        cons = [i for i in self.consumers.get(t, [])]
        if t not in self.graph_outs and len(cons) == 2:
            a, b = (self.nodes[i] for i in cons)
            for s, mnode, si, mi in ((a, b, cons[0], cons[1]), (b, a, cons[1], cons[0])):
                if s["op"] == "Sigmoid" and mnode["op"] == "Mul" and sorted(mnode["inputs"]) == sorted([t, s["outputs"][0]]) and \
                        self.consumers.get(s["outputs"][0], []) == [mi]:
                    used.update((si, mi))
                    return M.ACT_SILU, mnode["outputs"][0]
        i = self._single_consumer(t)
        if i is not None:
            nd = self.nodes[i]
            if nd["op"] == "Relu":
                used.add(i)
                return M.ACT_RELU, nd["outputs"][0]
            if nd["op"] == "LeakyRelu" and abs(float(nd["attrs"].get("alpha", 0.01)) - 0.1) < 1e-6:
                used.add(i)
                return M.ACT_LEAKY, nd["outputs"][0]
            # (HardSwish / HardSigmoid / Clip(0, 6) behind a convolution: the conv keeps ACT_NONE, the node becomes its own element-wise layer)
        return M.ACT_NONE, t

    def _macro_ops(self):
        """[(kind, node index, dict)] in graph order; activation / residual nodes absorbed into their convolution."""
        used, ops = set(), []
        for i, nd in enumerate(self.nodes):
            if i in used:
                continue
            op = nd["op"]
            if op in ("Conv", "ConvTranspose"):
                w = _const(self.m, nd["inputs"][1]) if len(nd["inputs"]) > 1 else None
                if w is None:
                    raise LowerError("node %s: convolution weights must be an initializer" % nd["name"])
                rec = dict(x=nd["inputs"][0], w=np.asarray(w, np.float32),
                           b=np.asarray(_const(self.m, nd["inputs"][2]), np.float32) if len(nd["inputs"]) > 2 and nd["inputs"][2] else None,
                           name=self._layer_name(nd), res=None, res_mode=M.RES_NONE)
                t = nd["outputs"][0]
                if op == "Conv":      # Conv -> BatchNormalization (an export that did not fuse them): folded into the weights
                    jb = self._single_consumer(t, "BatchNormalization")
                    if jb is not None:
                        bn = self.nodes[jb]
                        cs = [_const(self.m, x) for x in bn["inputs"][1:5]]
                        if len(cs) == 4 and all(c is not None for c in cs):
                            gam, bet, mu, var = (np.asarray(c, np.float64).reshape(-1) for c in cs)
                            eps = float(bn["attrs"].get("epsilon", 1e-5))
                            sc_ = gam / np.sqrt(var + eps)
                            b0 = rec["b"].astype(np.float64) if rec["b"] is not None else np.zeros(rec["w"].shape[0])
                            rec["w"] = (rec["w"].astype(np.float64) * sc_.reshape(-1, 1, 1, 1)).astype(np.float32)
                            rec["b"] = ((b0 - mu) * sc_ + bet).astype(np.float32)
                            used.add(jb)
                            t = bn["outputs"][0]
                if op == "ConvTranspose":
                    rec.update(act=M.ACT_NONE, out=t)
                    ops.append(("deconv", i, rec))
                    continue
                # Conv -> Add(residual) -> act  (ResNet)   |   Conv -> act -> Add(residual)  (Bottleneck shortcut)
                j = self._single_consumer(t, "Add")
                if j is not None and self._is_act_input(self.nodes[j]["outputs"][0]):
                    other = [x for x in self.nodes[j]["inputs"] if x != t]
                    if len(other) == 1 and other[0] not in self.m.initializers:
                        used.add(j)
                        act, out = self._absorb_act(self.nodes[j]["outputs"][0], used)
                        rec.update(act=act, out=out, res=other[0], res_mode=M.RES_BEFORE_ACT)
                        ops.append(("conv", i, rec))
                        continue
                act, out = self._absorb_act(t, used)
                j = self._single_consumer(out, "Add")
                if j is not None:
                    other = [x for x in self.nodes[j]["inputs"] if x != out]
                    if len(other) == 1 and other[0] not in self.m.initializers and self._defined_before(other[0], i):
                        used.add(j)
                        rec.update(res=other[0], res_mode=M.RES_AFTER_ACT)
                        out = self.nodes[j]["outputs"][0]
                rec.update(act=act, out=out)
                ops.append(("conv", i, rec))
            elif op == "Reshape" and self._match_shuffle(i) is not None:
                j1, j2, groups = self._match_shuffle(i)
                used.update((j1, j2))
                ops.append(("shuffle", i, dict(x=nd["inputs"][0], out=self.nodes[j2]["outputs"][0], groups=groups,
                                               name=(nd["name"].strip("/").replace("/", ".") or "shuffle%d" % i)[-47:])))
            elif op == "Reshape" and self._looks_like_attention(nd):
                rec = self._match_attention(i, used)
                ops.append(("attn", i, rec) if rec is not None else ("other", i, {}))
            elif op == "GlobalAveragePool":
                rec = self._match_se(i, used)
                ops.append(("se", i, rec) if rec is not None else ("other", i, {}))
            elif op == "Clip" and self._is_relu6(nd):
                ops.append(("wsum", i, dict(terms=[(nd["inputs"][0], 1.0, False)], act=M.ACT_RELU6, out=nd["outputs"][0],
                                            name=(nd["name"].strip("/").replace("/", ".") or "act%d" % i)[-47:])))
            elif op == "HardSwish" or (op == "HardSigmoid" and self._is_torch_hardsigmoid(nd)):
                # torch.nn.Hardswish / Hardsigmoid: a stand-alone activation layer = a ONE-input weighted sum (fuse_ops.hip wsum_kernel; the conv
                # epilogues do not carry these).  Exports below opset 14 write hard-swish as x * HardSigmoid(x).
                x, out, act = nd["inputs"][0], nd["outputs"][0], M.ACT_HSWISH if op == "HardSwish" else M.ACT_HSIGMOID
                if op == "HardSigmoid":
                    j = self._single_consumer(out, "Mul")
                    if j is not None and sorted(self.nodes[j]["inputs"]) == sorted([x, out]):
                        used.add(j)
                        act, out = M.ACT_HSWISH, self.nodes[j]["outputs"][0]
                ops.append(("wsum", i, dict(terms=[(x, 1.0, False)], act=act, out=out, name=(nd["name"].strip("/").replace("/", ".") or "act%d" % i)[-47:])))
            elif op == "Add" and self._is_free_sum(nd):
                # a stand-alone sum of feature maps (BiFPN fusion node, CBFuse, an identity shortcut around a block): the chain of Adds,
                # the constant scale in front of a term (Mul by a scalar initializer), a nearest x2 Resize that only this sum reads and the
                # activation behind it collapse into ONE weighted-sum op
                terms = self._sum_terms(nd["outputs"][0], used, i)
                if terms is None or not 2 <= len(terms) <= 3:
                    ops.append(("other", i, {}))
                    continue
                act, out = self._absorb_act(nd["outputs"][0], used)
                ops.append(("wsum", i, dict(terms=terms, act=act, out=out, name=(nd["name"].strip("/").replace("/", ".") or "sum%d" % i)[-47:])))
            elif op in ("Mul", "Resize", "Upsample") and self._feeds_only_free_sum(nd):
                continue          # folded into the weighted sum that consumes it (emitted there)
            elif op in ("MaxPool", "AveragePool", "Resize", "Upsample", "Concat", "Split", "Slice", "Identity"):
                ops.append((op.lower(), i, {}))
            else:
                ops.append(("other", i, {}))
        return ops

    # ---- channel shuffle
    def _match_shuffle(self, i):
        """Reshape (B, g, C / g, H, W) -> Transpose (0, 2, 1, 3, 4) -> Reshape (B, C, H, W): torch channel_shuffle.  -> (transpose node, second
        reshape node, groups) or None."""
        nd = self.nodes[i]
        shp = _ints(self.m, nd, "shape", 1)
        if shp is None or len(shp) != 5:
            return None
        j1 = self._single_consumer(nd["outputs"][0], "Transpose")
        if j1 is None or _ints(self.m, self.nodes[j1], "perm", None) != [0, 2, 1, 3, 4]:
            return None
        j2 = self._single_consumer(self.nodes[j1]["outputs"][0], "Reshape")
        if j2 is None:
            return None
        shp2 = _ints(self.m, self.nodes[j2], "shape", 1)
        if shp2 is None or len(shp2) != 4 or shp2[1] not in (-1, shp[1] * shp[2]):
            return None
        return j1, j2, int(shp[1])

    # ---- softmax attention (ultralytics Attention: YOLOv10 PSA)
    def _looks_like_attention(self, nd):
        j = self._single_consumer(nd["outputs"][0], "Split")
        return j is not None and len(self.nodes[j]["outputs"]) == 3 and int(self.nodes[j]["attrs"].get("axis", 0)) == 2

    def _match_attention(self, i, used):
        """qkv -> Reshape (B, heads, 2 kd + hd, N) -> Split(q, k, v) -> Softmax((q^T k) * scale) -> v attn^T -> Reshape (B, C, H, W), plus the
        depth-wise `pe` convolution of v.reshape(B, C, H, W), Add: one OP_ATTENTION and one depth-wise conv per head (each adds its slice of
        the attention output).  None when the nodes do not form exactly this pattern."""
        m, nd = self.m, self.nodes[i]
        one = lambda t, op: self._single_consumer(t, op)
        shp = _ints(m, nd, "shape", 1)
        js = one(nd["outputs"][0], "Split")
        sp = self.nodes[js]
        parts = _ints(m, sp, "split", 1)
        if shp is None or len(shp) != 4 or parts is None or len(parts) != 3 or parts[0] != parts[1] or sum(parts) != shp[2]:
            return None
        nh, kd, hd = int(shp[1]), int(parts[0]), int(parts[2])
        q, k, v = sp["outputs"]
        jt = one(q, "Transpose")
        if jt is None or _ints(m, self.nodes[jt], "perm", None) != [0, 1, 3, 2]:
            return None
        jm = one(self.nodes[jt]["outputs"][0], "MatMul")
        if jm is None or self.nodes[jm]["inputs"] != [self.nodes[jt]["outputs"][0], k] or self.consumers.get(k, []) != [jm]:
            return None
        t, chain, scale = self.nodes[jm]["outputs"][0], [jt, jm], 1.0
        jx = one(t, "Mul")
        if jx is not None:
            cs = [x for x in self.nodes[jx]["inputs"] if x in m.initializers and np.asarray(m.initializers[x]).size == 1]
            if len(cs) != 1:
                return None
            scale = float(np.asarray(m.initializers[cs[0]], np.float32).reshape(-1)[0])
            chain.append(jx)
            t = self.nodes[jx]["outputs"][0]
        jsm = one(t, "Softmax")
        if jsm is None or int(self.nodes[jsm]["attrs"].get("axis", -1)) not in (-1, 3):
            return None
        jt2 = one(self.nodes[jsm]["outputs"][0], "Transpose")
        if jt2 is None or _ints(m, self.nodes[jt2], "perm", None) != [0, 1, 3, 2]:
            return None
        jm2 = one(self.nodes[jt2]["outputs"][0], "MatMul")
        if jm2 is None or self.nodes[jm2]["inputs"] != [v, self.nodes[jt2]["outputs"][0]]:
            return None
        jr = one(self.nodes[jm2]["outputs"][0], "Reshape")
        vc = [c for c in self.consumers.get(v, []) if c != jm2]
        if jr is None or len(vc) != 1 or self.nodes[vc[0]]["op"] != "Reshape":
            return None
        jc = one(self.nodes[vc[0]]["outputs"][0], "Conv")
        if jc is None:
            return None
        pe = self.nodes[jc]
        pw = _const(m, pe["inputs"][1]) if len(pe["inputs"]) > 1 else None
        pb = _const(m, pe["inputs"][2]) if len(pe["inputs"]) > 2 and pe["inputs"][2] else None
        C = nh * hd
        if pw is None or int(pe["attrs"].get("group", 1)) != C or tuple(np.asarray(pw).shape[:2]) != (C, 1) or np.asarray(pw).shape[2] != np.asarray(pw).shape[3]:
            return None
        kpe = int(np.asarray(pw).shape[2])
        if (_ints(m, pe, "pads", None) or [0] * 4) != [kpe // 2] * 4 or (_ints(m, pe, "strides", None) or [1, 1]) != [1, 1]:
            return None
        ja = one(pe["outputs"][0], "Add")
        o4 = self.nodes[jr]["outputs"][0]
        if ja is None or sorted(self.nodes[ja]["inputs"]) != sorted([o4, pe["outputs"][0]]) or self.consumers.get(o4, []) != [ja]:
            return None
        if abs(scale - float(kd) ** -0.5) > 1e-6 * max(1.0, abs(scale)):
            raise LowerError("node %s: attention scale %g is not key_dim ** -0.5 = %g" % (nd["name"], scale, float(kd) ** -0.5))
        used.update(chain + [js, jsm, jt2, jm2, jr, vc[0], jc, ja])
        src = self.nodes[self.producer[nd["inputs"][0]]] if nd["inputs"][0] in self.producer else None
        base = self._layer_name(src) if src is not None and src["op"] == "Conv" else "attn%d" % i
        base = base.rsplit(".qkv", 1)[0] if ".qkv" in base else base
        return dict(x=nd["inputs"][0], out=self.nodes[ja]["outputs"][0], nh=nh, kd=kd, hd=hd, N=int(shp[3]), name=base[-30:],
                    pw=np.asarray(pw, np.float32), pb=np.asarray(pb, np.float32) if pb is not None else np.zeros(C, np.float32))

    # ---- squeeze-and-excitation
    def _match_se(self, i, used):
        """GlobalAveragePool(x) -> Conv 1x1 -> swish -> Conv 1x1 -> Sigmoid -> Mul(x, .): one gate + scale pair (OP_SE_GATE, OP_SCALE).  Returns
        None when the nodes behind the pool are anything else (the pool is then reported as unsupported)."""
        nd = self.nodes[i]
        x, t = nd["inputs"][0], nd["outputs"][0]
        j1 = self._single_consumer(t, "Conv")
        if j1 is None:
            return None
        c1 = self.nodes[j1]
        w1 = _const(self.m, c1["inputs"][1]) if len(c1["inputs"]) > 1 else None
        b1 = _const(self.m, c1["inputs"][2]) if len(c1["inputs"]) > 2 and c1["inputs"][2] else None
        if w1 is None or np.asarray(w1).ndim != 4 or tuple(np.asarray(w1).shape[2:]) != (1, 1):
            return None
        tmp = set()
        act, a = self._absorb_act(c1["outputs"][0], tmp)
        if act not in (M.ACT_SILU, M.ACT_RELU):      # EfficientNet: swish; MobileNetV3 / PP-LCNet: ReLU
            return None
        j2 = self._single_consumer(a, "Conv")
        if j2 is None:
            return None
        c2 = self.nodes[j2]
        w2 = _const(self.m, c2["inputs"][1]) if len(c2["inputs"]) > 1 else None
        b2 = _const(self.m, c2["inputs"][2]) if len(c2["inputs"]) > 2 and c2["inputs"][2] else None
        if w2 is None or tuple(np.asarray(w2).shape[2:]) != (1, 1) or np.asarray(w2).shape[:2] != np.asarray(w1).shape[1::-1]:
            return None
        j3, gate_act = self._single_consumer(c2["outputs"][0], "Sigmoid"), M.ACT_NONE
        if j3 is None:
            j3, gate_act = self._single_consumer(c2["outputs"][0], "HardSigmoid"), M.ACT_HSIGMOID
            if j3 is None or not self._is_torch_hardsigmoid(self.nodes[j3]):
                return None
        j4 = self._single_consumer(self.nodes[j3]["outputs"][0], "Mul")
        if j4 is None or sorted(self.nodes[j4]["inputs"]) != sorted([x, self.nodes[j3]["outputs"][0]]):
            return None
        used.update(tmp)
        used.update((j1, j2, j3, j4))
        w1, w2 = np.asarray(w1, np.float32), np.asarray(w2, np.float32)
        name = self._layer_name(c1)
        name = name[:-len(".reduce")] if name.endswith(".reduce") else name
        return dict(x=x, out=self.nodes[j4]["outputs"][0], name=name[-40:], w1=w1, w2=w2, hidden_act=act, gate_act=gate_act,
                    b1=np.asarray(b1, np.float32) if b1 is not None else np.zeros(w1.shape[0], np.float32),
                    b2=np.asarray(b2, np.float32) if b2 is not None else np.zeros(w2.shape[0], np.float32))

    @staticmethod
    def _is_torch_hardsigmoid(nd):
        """HardSigmoid(alpha = 1/6, beta = 0.5) = relu6(x + 3) / 6, torch.nn.Hardsigmoid (ONNX's default alpha is 0.2: another function)."""
        return abs(float(nd["attrs"].get("alpha", 0.2)) - 1.0 / 6.0) < 1e-6 and abs(float(nd["attrs"].get("beta", 0.5)) - 0.5) < 1e-6

    def _is_relu6(self, nd):
        """Clip(x, 0, 6) = torch.nn.ReLU6: bounds as inputs 1, 2 (opset >= 11) or as the min / max attributes."""
        lo = _const(self.m, nd["inputs"][1]) if len(nd["inputs"]) > 1 and nd["inputs"][1] else nd["attrs"].get("min")
        hi = _const(self.m, nd["inputs"][2]) if len(nd["inputs"]) > 2 and nd["inputs"][2] else nd["attrs"].get("max")
        if lo is None or hi is None:
            return False
        return float(np.asarray(lo).reshape(-1)[0]) == 0.0 and float(np.asarray(hi).reshape(-1)[0]) == 6.0

    # ---- stand-alone sums
    def _is_free_sum(self, nd):
        """An Add of two non-constant tensors that is the END of its chain (its output does not feed another Add of the same kind through a
        single-consumer link) and is not a convolution's residual (those are absorbed before this node is reached)."""
        if len(nd["inputs"]) != 2 or any(x in self.m.initializers for x in nd["inputs"]):
            return False
        j = self._single_consumer(nd["outputs"][0], "Add")
        if j is not None and not any(x in self.m.initializers for x in self.nodes[j]["inputs"]):
            return False          # an inner link of a longer chain: handled from the chain's last Add
        return True

    def _is_nearest_x2(self, nd):
        """Resize / Upsample by exactly (1, 1, 2, 2), nearest: the only up-sampling the engine has (and the only one a sum may fold)."""
        m = self.m
        mode = nd["attrs"].get("mode", b"nearest")
        mode = mode.decode() if isinstance(mode, (bytes, bytearray)) else str(mode)
        if mode != "nearest":
            return False
        if isinstance(nd["attrs"].get("scales"), list) and len(nd["attrs"]["scales"]) == 4:     # Upsample, opset 7
            return [float(x) for x in nd["attrs"]["scales"]] == [1.0, 1.0, 2.0, 2.0]
        for idx in (2, 1):
            if len(nd["inputs"]) > idx and nd["inputs"][idx] and _const(m, nd["inputs"][idx]) is not None and np.asarray(_const(m, nd["inputs"][idx])).size == 4:
                return [float(x) for x in np.asarray(_const(m, nd["inputs"][idx])).reshape(-1)] == [1.0, 1.0, 2.0, 2.0]
        return False          # sizes-based Resize: left to the stand-alone op's own shape check

    def _term(self, t, used):
        """(source tensor, weight, upsampled) of one term: through Mul(x, scalar constant) and a nearest x2 Resize read by nobody else."""
        wgt, up = 1.0, False
        for _ in range(3):
            i = self.producer.get(t)
            if i is None or len(self.consumers.get(t, [])) != 1 or t in self.graph_outs:
                break
            nd = self.nodes[i]
            if nd["op"] == "Mul":
                cs = [x for x in nd["inputs"] if x in self.m.initializers and np.asarray(self.m.initializers[x]).size == 1]
                xs = [x for x in nd["inputs"] if x not in self.m.initializers]
                if len(cs) != 1 or len(xs) != 1:
                    break
                wgt *= float(np.asarray(self.m.initializers[cs[0]], np.float32).reshape(-1)[0])
                used.add(i)
                t = xs[0]
            elif nd["op"] in ("Resize", "Upsample") and not up and self._is_nearest_x2(nd):
                used.add(i)
                up = True
                t = nd["inputs"][0]
            else:
                break
        return t, wgt, up

    def _sum_terms(self, t, used, last):
        """Terms of the Add chain ending in tensor t, left to right; inner Adds are marked used."""
        nd = self.nodes[self.producer[t]]
        terms = []
        for x in nd["inputs"]:
            i = self.producer.get(x)
            if i is not None and self.nodes[i]["op"] == "Add" and len(self.consumers.get(x, [])) == 1 and x not in self.graph_outs and \
                    not any(y in self.m.initializers for y in self.nodes[i]["inputs"]) and i not in used:
                inner = self._sum_terms(x, used, last)
                if inner is None:
                    return None
                used.add(i)
                terms += inner
            else:
                terms.append(self._term(x, used))
        return terms

    def _feeds_only_free_sum(self, nd):
        """Mul-by-constant / Resize nodes in front of a stand-alone sum are visited BEFORE the Add (graph order): decide here, without the
        `used` set, whether the sum will swallow them."""
        if nd["op"] in ("Resize", "Upsample") and not self._is_nearest_x2(nd):
            return False          # stays a stand-alone op (and fails there, by name, if the engine has no kernel for it)
        if nd["op"] == "Mul":     # only a scale by a scalar constant is a term's weight (x * sigmoid(x), x * gate are not)
            cs = [x for x in nd["inputs"] if x in self.m.initializers and np.asarray(self.m.initializers[x]).size == 1]
            if len(cs) != 1 or len(nd["inputs"]) != 2:
                return False
        t = nd["outputs"][0]
        for _ in range(3):
            c = self.consumers.get(t, [])
            if len(c) != 1 or t in self.graph_outs:
                return False
            nx = self.nodes[c[0]]
            if nx["op"] == "Add":
                if any(x in self.m.initializers for x in nx["inputs"]):
                    return False
                # walk to the end of the chain and ask whether THAT add is a free sum whose term count fits
                end = nx
                while True:
                    j = self._single_consumer(end["outputs"][0], "Add")
                    if j is None or any(x in self.m.initializers for x in self.nodes[j]["inputs"]):
                        break
                    end = self.nodes[j]
                if not self._is_free_sum(end):
                    return False
                n_terms = self._count_terms(end["outputs"][0])
                return 2 <= n_terms <= 3 and self._is_sum_not_residual(end)
            if nx["op"] == "Mul" and nd["op"] != "Mul":
                cs = [x for x in nx["inputs"] if x in self.m.initializers and np.asarray(self.m.initializers[x]).size == 1]
                if len(cs) != 1:
                    return False
                t = nx["outputs"][0]
                continue
            return False
        return False

    def _count_terms(self, t):
        nd = self.nodes[self.producer[t]]
        n = 0
        for x in nd["inputs"]:
            i = self.producer.get(x)
            if i is not None and self.nodes[i]["op"] == "Add" and len(self.consumers.get(x, [])) == 1 and x not in self.graph_outs and \
                    not any(y in self.m.initializers for y in self.nodes[i]["inputs"]):
                n += self._count_terms(x)
            else:
                n += 1
        return n

    def _is_sum_not_residual(self, end):
        """False when the chain's last Add is one a convolution absorbs as its residual (Conv -> Add, Conv -> act -> Add)."""
        for x in end["inputs"]:
            i = self.producer.get(x)
            if i is None:
                continue
            src = self.nodes[i]
            if src["op"] == "Conv" and len(self.consumers.get(x, [])) == 1:
                return False
            if src["op"] in ("Relu", "LeakyRelu", "Mul") and len(self.consumers.get(x, [])) == 1:
                # activation of a convolution?  (Mul = x * sigmoid(x))
                y = src["inputs"][0]
                k = self.producer.get(y)
                if k is not None and self.nodes[k]["op"] == "Conv":
                    return False
        return True

    def _is_act_input(self, t):
        c = self.consumers.get(t, [])
        return len(c) == 1 and self.nodes[c[0]]["op"] in ("Relu", "LeakyRelu") or \
            (len(c) == 2 and {self.nodes[k]["op"] for k in c} == {"Sigmoid", "Mul"})

    def _defined_before(self, t, node_idx):
        while t in self.producer and self.nodes[self.producer[t]]["op"] in ("Slice", "Split", "Identity"):   # views: what matters is their source
            t = self.nodes[self.producer[t]]["inputs"][0]
        return t == self.in_name or self.producer.get(t, 1 << 30) < node_idx

    def _layer_name(self, nd):
        w = nd["inputs"][1] if len(nd["inputs"]) > 1 else ""
        base = w[:-len(".weight")] if w.endswith(".weight") else (nd["name"].strip("/").replace("/", ".") or w)
        return base[-47:] if base else "conv%d" % self.producer[nd["outputs"][0]]

    # ------------------------------------------------------------------ lowering proper
    def run(self):
        m = self.m
        ops = self._macro_ops()
        # ---- the Detect tail: found first, so that everything behind its front is skipped
        tail = self._find_tail(ops)
        # the network body = the ancestors of the Detect inputs (the tail's own nodes, and whatever only feeds them, are the Detect op)
        made_by = {}
        for kind, i, rec in ops:
            if kind in ("conv", "deconv", "wsum", "se", "attn", "shuffle"):
                made_by[rec["out"]] = (kind, i, rec)
            else:
                for o in self.nodes[i]["outputs"]:
                    made_by[o] = (kind, i, rec)
        need, todo = set(), list(tail["f32_tensors"])
        while todo:
            t = todo.pop()
            if t == self.in_name or t not in made_by:
                if t != self.in_name:
                    raise LowerError("tensor %r feeding the Detect head has no producer among the supported nodes" % t)
                continue
            kind, i, rec = made_by[t]
            if i in need:
                continue
            need.add(i)
            srcs = [rec["x"]] + ([rec["res"]] if rec.get("res") is not None else []) if kind in ("conv", "deconv") else \
                [t_ for t_, _, _ in rec["terms"]] if kind == "wsum" else [rec["x"]] if kind in ("se", "attn", "shuffle") else \
                [x for x in self.nodes[i]["inputs"] if x and x not in m.initializers]
            todo += srcs
        body = [o for o in ops if o[1] in need]
        # ---- pass 1b: shapes, aliases (Split / Slice views), concat placement
        alias = {}                                   # tensor -> (source tensor, channel offset)
        place = {}                                   # tensor -> (concat output tensor, channel offset)
        copies = []                                  # (concat tensor, offset, source tensor, src offset, channels)
        root_of = lambda t: self._root(t, alias)

        for kind, i, rec in body:
            nd = self.nodes[i]
            if kind == "conv":
                c, h, w_ = self._shape(rec["x"])
                W = rec["w"]
                group = int(nd["attrs"].get("group", 1))
                k = W.shape[2]
                if W.shape[2] != W.shape[3]:
                    raise LowerError("node %s: %dx%d kernel (square kernels only)" % (nd["name"], W.shape[2], W.shape[3]))
                st = _ints(m, nd, "strides", None) or [1, 1]
                pd = _ints(m, nd, "pads", None) or [0, 0, 0, 0]
                dl = _ints(m, nd, "dilations", None) or [1, 1]
                ap = nd["attrs"].get("auto_pad")
                ap = ap.decode() if isinstance(ap, (bytes, bytearray)) else ap
                if ap in ("SAME_UPPER", "SAME_LOWER"):
                    if st != [1, 1] or k % 2 == 0:
                        raise LowerError("node %s: auto_pad %s with stride %s / kernel %d needs asymmetric padding (symmetric convolutions only)" % (nd["name"], ap, st, k))
                    pd = [k // 2] * 4
                elif ap not in (None, "NOTSET", "VALID"):
                    raise LowerError("node %s: auto_pad %s" % (nd["name"], ap))
                if st[0] != st[1] or len(set(pd)) != 1 or dl != [1, 1]:
                    raise LowerError("node %s: strides %s pads %s dilations %s (symmetric, undilated convolutions only)" % (nd["name"], st, pd, dl))
                if group not in (1, c) or (group == c and (W.shape[0] != c or c == 1)):
                    if group != 1:
                        raise LowerError("node %s: group %d of %d channels (plain and depth-wise convolutions only)" % (nd["name"], group, c))
                if group == 1 and W.shape[1] != c:
                    raise LowerError("node %s: weight expects %d input channels, tensor has %d" % (nd["name"], W.shape[1], c))
                rec.update(k=k, s=st[0], p=pd[0], dw=(group == c and group > 1))
                ho, wo = (h + 2 * pd[0] - k) // st[0] + 1, (w_ + 2 * pd[0] - k) // st[0] + 1
                self.shape[rec["out"]] = (W.shape[0], ho, wo)
                if rec["res"] is not None and self._shape(rec["res"]) != self.shape[rec["out"]]:
                    raise LowerError("node %s: residual shape %s != output shape %s" % (nd["name"], self._shape(rec["res"]), self.shape[rec["out"]]))
            elif kind == "deconv":
                c, h, w_ = self._shape(rec["x"])
                W = rec["w"]
                st = _ints(m, nd, "strides", None) or [1, 1]
                if tuple(W.shape[2:]) != (2, 2) or st != [2, 2] or W.shape[0] != c or int(nd["attrs"].get("group", 1)) != 1:
                    raise LowerError("node %s: only ConvTranspose2d(kernel 2, stride 2) is built" % nd["name"])
                self.shape[rec["out"]] = (W.shape[1], 2 * h, 2 * w_)
            elif kind == "shuffle":
                c, h, w_ = self._shape(rec["x"])
                if c % rec["groups"] or c % 8:
                    raise LowerError("node %s: channel shuffle of %d channels in %d groups (multiples of 8 only)" % (nd["name"], c, rec["groups"]))
                self.shape[rec["out"]] = (c, h, w_)
            elif kind == "attn":
                c, h, w_ = self._shape(rec["x"])
                if c != rec["nh"] * (2 * rec["kd"] + rec["hd"]) or h * w_ != rec["N"]:
                    raise LowerError("node %s: attention over a %s tensor with %d heads of %d + %d + %d channels and %d tokens" % (
                        nd["name"], (c, h, w_), rec["nh"], rec["kd"], rec["kd"], rec["hd"], rec["N"]))
                self.shape[rec["out"]] = (rec["nh"] * rec["hd"], h, w_)
            elif kind == "se":
                c, h, w_ = self._shape(rec["x"])
                if c % 8 or rec["w1"].shape[1] != c:
                    raise LowerError("node %s: squeeze-and-excitation over %d channels (weights for %d; multiples of 8 only)" % (nd["name"], c, rec["w1"].shape[1]))
                self.shape[rec["out"]] = (c, h, w_)
            elif kind == "wsum":
                shp = []
                for t_, _, up in rec["terms"]:
                    c, h, w_ = self._shape(t_)
                    shp.append((c, 2 * h, 2 * w_) if up else (c, h, w_))
                if len(set(shp)) != 1 or shp[0][0] % 8:
                    raise LowerError("node %s: sum of maps of shapes %s (one shape, channels a multiple of 8)" % (nd["name"], shp))
                self.shape[rec["out"]] = shp[0]
            elif kind in ("maxpool", "averagepool"):
                c, h, w_ = self._shape(nd["inputs"][0])
                ks = _ints(m, nd, "kernel_shape", None)
                st = _ints(m, nd, "strides", None) or [1, 1]
                pd = _ints(m, nd, "pads", None) or [0, 0, 0, 0]
                if ks[0] != ks[1] or st[0] != st[1] or len(set(pd)) != 1 or int(nd["attrs"].get("ceil_mode", 0)):
                    raise LowerError("node %s: pooling %s / %s / %s" % (nd["name"], ks, st, pd))
                if kind == "averagepool" and pd[0] and not int(nd["attrs"].get("count_include_pad", 0)):
                    raise LowerError("node %s: AveragePool with padding needs count_include_pad=1" % nd["name"])
                self.shape[nd["outputs"][0]] = (c, (h + 2 * pd[0] - ks[0]) // st[0] + 1, (w_ + 2 * pd[0] - ks[0]) // st[0] + 1)
            elif kind in ("resize", "upsample"):
                c, h, w_ = self._shape(nd["inputs"][0])
                mode = nd["attrs"].get("mode", b"nearest")
                mode = mode.decode() if isinstance(mode, (bytes, bytearray)) else str(mode)
                sc = [float(x) for x in nd["attrs"]["scales"]] if isinstance(nd["attrs"].get("scales"), list) and len(nd["attrs"]["scales"]) == 4 else None
                for idx in ((2, 1) if sc is None else ()):
                    if len(nd["inputs"]) > idx and nd["inputs"][idx] and _const(m, nd["inputs"][idx]) is not None and np.asarray(_const(m, nd["inputs"][idx])).size == 4:
                        sc = [float(x) for x in np.asarray(_const(m, nd["inputs"][idx])).reshape(-1)]
                        break
                if sc is None and len(nd["inputs"]) > 3 and nd["inputs"][3] and _const(m, nd["inputs"][3]) is not None:
                    sz = [int(x) for x in np.asarray(_const(m, nd["inputs"][3])).reshape(-1)]
                    sc = [1.0, 1.0, sz[2] / h, sz[3] / w_]
                if mode != "nearest" or sc is None or sc[:2] != [1.0, 1.0] or sc[2] != sc[3] or sc[2] not in (2.0, 4.0, 8.0):
                    raise LowerError("node %s: only nearest-neighbour x2 / x4 / x8 up-sampling is built (mode %s, scales %s)" % (nd["name"], mode, sc))
                rec["factor"] = int(sc[2])       # x4 / x8 (YOLOv9 CBFuse): a chain of x2 launches
                self.shape[nd["outputs"][0]] = (c, int(sc[2]) * h, int(sc[2]) * w_)
            elif kind == "identity":
                self.shape[nd["outputs"][0]] = self._shape(nd["inputs"][0])
                alias[nd["outputs"][0]] = (nd["inputs"][0], 0)
            elif kind == "split":
                c, h, w_ = self._shape(nd["inputs"][0])
                axis = int(nd["attrs"].get("axis", 0))
                parts = _ints(m, nd, "split", 1) or [c // len(nd["outputs"])] * len(nd["outputs"])
                if axis != 1 or sum(parts) != c:
                    raise LowerError("node %s: Split along axis %d / %s of %d channels" % (nd["name"], axis, parts, c))
                off = 0
                for o, pc in zip(nd["outputs"], parts):
                    self.shape[o] = (pc, h, w_)
                    alias[o] = (nd["inputs"][0], off)
                    off += pc
            elif kind == "slice":
                c, h, w_ = self._shape(nd["inputs"][0])
                starts, ends = _ints(m, nd, "starts", 1), _ints(m, nd, "ends", 2)
                axes = _ints(m, nd, "axes", 3) or [0]
                steps = _ints(m, nd, "steps", 4) or [1]
                if axes != [1] or steps != [1] or len(starts) != 1:
                    raise LowerError("node %s: Slice on axes %s steps %s (channel slices only)" % (nd["name"], axes, steps))
                s0 = starts[0] + c if starts[0] < 0 else starts[0]
                e0 = min(c, ends[0] + c if ends[0] < 0 else ends[0])
                self.shape[nd["outputs"][0]] = (e0 - s0, h, w_)
                alias[nd["outputs"][0]] = (nd["inputs"][0], s0)
            elif kind == "concat":
                if int(nd["attrs"].get("axis", 0)) != 1:
                    raise LowerError("node %s: Concat along axis %s in the network body" % (nd["name"], nd["attrs"].get("axis")))
                shp = [self._shape(t) for t in nd["inputs"]]
                if len({s[1:] for s in shp}) != 1:
                    raise LowerError("node %s: Concat of different spatial sizes %s" % (nd["name"], shp))
                self.shape[nd["outputs"][0]] = (sum(s[0] for s in shp), shp[0][1], shp[0][2])
            else:
                raise LowerError("node %s: op %s has no counterpart in the engine (element-wise arithmetic other than sums of 2-3 feature maps, swish "
                                 "squeeze-and-excitation gates and ultralytics-style softmax attention needs a hand-written builder)" % (nd["name"] or "#%d" % i, nd["op"]))

        # concat placement, in graph order
        for kind, i, rec in body:
            if kind != "concat":
                continue
            nd = self.nodes[i]
            out, off = nd["outputs"][0], 0
            k = 0
            ins = nd["inputs"]
            while k < len(ins):
                r, ro = root_of(ins[k])
                c = self.shape[ins[k]][0]
                # a run of consecutive slices of one source tensor that covers it from channel 0 in order places the whole source
                run_c, k2 = c, k + 1
                if ro == 0:
                    while k2 < len(ins) and run_c < self.shape[r][0]:
                        r2, ro2 = root_of(ins[k2])
                        if r2 != r or ro2 != run_c:
                            break
                        run_c += self.shape[ins[k2]][0]
                        k2 += 1
                whole = ro == 0 and run_c == self.shape[r][0]
                if whole and r not in place and r != self.in_name and r not in self.graph_outs:
                    place[r] = (out, off)
                    off += run_c
                    k = k2
                else:
                    copies.append((out, off, r, ro, c))
                    off += c
                    k += 1
        self.alias, self.place = alias, place

        # ---- pass 2: emission
        g = M.Graph(self.name, self.in_c, self.in_h, self.in_w, M.DictWeights({}))
        x_in, _ = g.input()
        self.g, self.bufs = g, {}
        self.bufs[self.in_name] = x_in.buf
        f32 = set(tail["f32_tensors"])
        self.f32 = f32
        first = True
        copy_at = {}
        for cp in copies:
            copy_at.setdefault(self.producer.get(cp[0]), []).append(cp)
        for kind, i, rec in body:
            nd = self.nodes[i]
            if kind == "conv":
                xin = self._view(rec["x"])
                out = self._view(rec["out"], make=True)
                res = self._view(rec["res"]) if rec["res"] is not None else None
                if rec["dw"]:
                    if rec["res_mode"] == M.RES_BEFORE_ACT:
                        raise LowerError("node %s: depth-wise convolution with a pre-activation residual" % nd["name"])
                    b = rec["b"] if rec["b"] is not None else np.zeros(rec["w"].shape[0], np.float32)
                    g.dwconv(xin, rec["k"], rec["s"], rec["name"], act=rec["act"], out=out, res=res, weight=rec["w"].reshape(-1, 1, rec["k"], rec["k"]), bias=b)
                    if rec["p"] != rec["k"] // 2:
                        raise LowerError("node %s: depth-wise padding %d (k // 2 only)" % (nd["name"], rec["p"]))
                else:
                    true_cin = self.in_c if rec["x"] == self.in_name else None
                    b = rec["b"] if rec["b"] is not None else np.zeros(rec["w"].shape[0], np.float32)
                    g.conv(xin, rec["w"].shape[0], rec["k"], rec["s"], rec["name"], act=rec["act"], out=out, res=res, res_mode=rec["res_mode"],
                           true_cin=true_cin, pad=rec["p"], weight=rec["w"], bias_arr=b)
                g.n_params += rec["w"].size + (rec["b"].size if rec["b"] is not None else 0)
                first = False
            elif kind == "shuffle":
                g.shuffle(self._view(rec["x"]), rec["groups"], rec["name"], out=self._view(rec["out"], make=True))
            elif kind == "attn":
                nh, kd, hd = rec["nh"], rec["kd"], rec["hd"]
                qkv = self._view(rec["x"])
                att = g.attention(qkv, nh, kd, hd, rec["name"] + ".softmax")
                summed = self._view(rec["out"], make=True)
                kpe = rec["pw"].shape[2]
                for h_ in range(nh):     # pe(v) per head, adding that head's slice of the attention output (models._psa_block)
                    v_ = qkv.slice(h_ * (2 * kd + hd) + 2 * kd, hd)
                    g.dwconv(v_, kpe, 1, "%s.pe.conv.h%d" % (rec["name"], h_), act=M.ACT_NONE, out=summed.slice(h_ * hd, hd), res=att.slice(h_ * hd, hd),
                             weight=rec["pw"][h_ * hd:(h_ + 1) * hd], bias=rec["pb"][h_ * hd:(h_ + 1) * hd])
                g.n_params += rec["pw"].size + rec["pb"].size
            elif kind == "se":
                n_ = rec["name"]
                g.w = M.DictWeights({n_ + ".reduce.weight": rec["w1"], n_ + ".reduce.bias": rec["b1"], n_ + ".expand.weight": rec["w2"], n_ + ".expand.bias": rec["b2"]})
                g.se(self._view(rec["x"]), rec["w1"].shape[0], n_, out=self._view(rec["out"], make=True), hidden_act=rec["hidden_act"], gate_act=rec["gate_act"])
            elif kind == "wsum":
                views = [self._view(t_) for t_, _, _ in rec["terms"]]
                g.wsum(views, [w_ for _, w_, _ in rec["terms"]], rec["name"], act=rec["act"], out=self._view(rec["out"], make=True))
            elif kind == "deconv":
                xin, out = self._view(rec["x"]), self._view(rec["out"], make=True)
                W, b = rec["w"], rec["b"] if rec["b"] is not None else np.zeros(rec["w"].shape[1], np.float32)
                g.w = M.DictWeights({rec["name"] + ".weight": W, rec["name"] + ".bias": b})
                g.deconv2x2(xin, W.shape[1], rec["name"], out=out)
            elif kind in ("maxpool", "averagepool"):
                ks = _ints(m, nd, "kernel_shape", None)[0]
                st = (_ints(m, nd, "strides", None) or [1, 1])[0]
                pd = (_ints(m, nd, "pads", None) or [0, 0, 0, 0])[0]
                xin, out = self._view(nd["inputs"][0]), self._view(nd["outputs"][0], make=True)
                (g.maxpool if kind == "maxpool" else g.avgpool)(xin, ks, st, pd, out=out, name=(nd["name"].strip("/").replace("/", ".") or kind)[-47:])
            elif kind in ("resize", "upsample"):
                nm = (nd["name"].strip("/").replace("/", ".") or "upsample")[-44:]
                v, f_ = self._view(nd["inputs"][0]), rec.get("factor", 2)
                while f_ > 2:                    # nearest x4 / x8 = nearest x2 applied two / three times
                    v = g.upsample2(v, name="%s.x%d" % (nm, f_))
                    f_ //= 2
                g.upsample2(v, out=self._view(nd["outputs"][0], make=True), name=nm)
            elif kind == "concat":
                self._view(nd["outputs"][0], make=True)
                for (ct, off, r, ro, c) in copy_at.get(i, []):
                    src = self._view(r).slice(ro, c)
                    dst = self._view(ct).slice(off, c)
                    g.maxpool(src, 1, 1, 0, out=dst, name=("copy.%s" % r)[-47:])           # a 1x1 max-pool moves the channels
        # ---- the Detect op
        self._emit_tail(tail)
        g.w = M.DictWeights({})
        return g

    # ------------------------------------------------------------------ helpers
    def _shape(self, t):
        if t not in self.shape:
            raise LowerError("tensor %r is used before any supported node defines it" % t)
        return self.shape[t]

    @staticmethod
    def _root(t, alias):
        off = 0
        while t in alias:
            t, o = alias[t]
            off += o
        return t, off

    def _physical(self, t):
        """(tensor that owns a buffer, channel offset inside it) of tensor t, through aliases and concat placements."""
        off = 0
        while True:
            if t in self.alias:
                t, o = self.alias[t]
                off += o
            elif t in self.place:
                t, o = self.place[t]
                off += o
            else:
                return t, off

    def _view(self, t, make=False):
        owner, off = self._physical(t)
        c, h, w_ = self._shape(t)
        if t == self.in_name:
            return M.View(self.bufs[t], 0, 8, h, w_)       # the engine's input layer: channels zero-padded to 8 (conv(..., true_cin))
        if owner not in self.bufs:
            if not make and owner != t:
                pass
            oc, oh, ow = self._shape(owner)
            self.bufs[owner] = self.g.buf(oh, ow, oc, f32=owner in self.f32).buf
        return M.View(self.bufs[owner], off, c, h, w_)

    # ------------------------------------------------------------------ Detect tails
    def _find_tail(self, ops):
        m = self.m
        if len(self.graph_outs) == 2:
            return self._find_raw_heads(ops)
        if len(self.graph_outs) != 1:
            raise LowerError("expected one graph output (a YOLO head) or two (box regression and class logits of an anchor-based head), found %s" % (self.graph_outs,))
        oshape = m.outputs[0][1]
        by_node = {i: (kind, rec) for kind, i, rec in ops}
        conv_out = {rec["out"]: (i, rec) for kind, i, rec in ops if kind == "conv"}
        # v8 layout: axis-2 Concat of three Reshape(Concat[box conv, cls conv])
        for i, nd in enumerate(self.nodes):
            if nd["op"] != "Concat" or int(nd["attrs"].get("axis", 0)) != 2 or len(nd["inputs"]) != 3:
                continue
            levels = []
            for t in nd["inputs"]:
                r = self.nodes[self.producer[t]] if t in self.producer else None
                if r is None or r["op"] != "Reshape":
                    break
                c = self.nodes[self.producer[r["inputs"][0]]] if r["inputs"][0] in self.producer else None
                if c is None or c["op"] != "Concat" or int(c["attrs"].get("axis", 0)) != 1 or len(c["inputs"]) != 2 or \
                        not all(x in conv_out and conv_out[x][1]["act"] == M.ACT_NONE and conv_out[x][1]["res"] is None for x in c["inputs"]):
                    break
                levels.append((self.producer[r["inputs"][0]], c["inputs"][0], c["inputs"][1]))
            if len(levels) != 3:
                continue
            box_c = conv_out[levels[0][1]][1]["w"].shape[0]
            nc = conv_out[levels[0][2]][1]["w"].shape[0]
            if box_c != 64 or len(oshape) != 3 or oshape[1] != 4 + nc:
                raise LowerError("Detect head with %d box channels / %d classes and output %s: only the reg_max = 16 v8 layout (1, 4 + nc, A) is built" % (box_c, nc, oshape))
            return dict(kind="v8", levels=[(b, c) for _, b, c in levels], nc=nc, out_name=self.graph_outs[0],
                        f32_tensors=[t for _, b, c in levels for t in (b, c)])
        # v5 layout: three 1x1 convs of 3 * no channels, each reshaped to (1, 3, no, h, w)
        if len(oshape) == 3:
            no = oshape[2]
            levels = []
            for kind, i, rec in ops:
                if kind != "conv" or rec["act"] != M.ACT_NONE or rec["res"] is not None or rec["w"].shape[0] != 3 * no or rec["w"].shape[2] != 1:
                    continue
                j = self._single_consumer(rec["out"], "Reshape")
                if j is None:
                    continue
                tgt = _ints(m, self.nodes[j], "shape", 1)
                if tgt is None or len(tgt) != 5 or tgt[1] != 3 or tgt[2] != no:
                    continue
                levels.append((j, rec["out"], self._anchors_behind(j)))
            if len(levels) == 3:
                return dict(kind="v5", levels=[(t, a) for _, t, a in levels], nc=no - 5, out_name=self.graph_outs[0],
                            f32_tensors=[t for _, t, _ in levels])
        raise LowerError("no Detect head recognised (output %s): the v8 layout needs three Concat[box, cls] -> Reshape pairs feeding an axis-2 Concat, "
                         "the v5 layout three 1x1 convolutions of 3 (5 + nc) channels reshaped to (1, 3, 5 + nc, h, w)" % (oshape,))

    def _find_raw_heads(self, ops):
        """A head-only export of an anchor-based detector (EfficientDet without its in-graph decode / NMS): two outputs, each the axis-1
        Concat over the pyramid levels of Reshape(Transpose(conv, NCHW -> NHWC), (1, -1, k)) -- box regression (k = 4) and class LOGITS
        (k = classes).  The convolutions' fp32 outputs ARE those tensors level by level (NHWC rows ordered (y, x, anchor)): the engine graph
        gets one output per level and tensor, named like the hand-built EfficientDet graph's (regression.l<i>, classification.l<i>), which
        coreEngine.EfficientdetEngine hands to the device tail."""
        conv_out = {rec["out"]: rec for kind, i, rec in ops if kind == "conv"}
        found = {}
        for oname, oshape in self.m.outputs:
            nd = self.nodes[self.producer[oname]] if oname in self.producer else None
            if nd is not None and nd["op"] == "Sigmoid":
                raise LowerError("output %s: class probabilities (a Sigmoid behind the head): export the logits, the device tail applies it" % oname)
            if nd is None or nd["op"] != "Concat" or int(nd["attrs"].get("axis", 0)) != 1 or len(oshape) != 3:
                raise LowerError("output %s %s: expected Concat(axis 1) of per-level Reshape(Transpose(conv)) tensors" % (oname, oshape))
            k, levels = int(oshape[2]), []
            for t in nd["inputs"]:
                r = self.nodes[self.producer[t]] if t in self.producer else None
                tr = self.nodes[self.producer[r["inputs"][0]]] if r is not None and r["op"] == "Reshape" and r["inputs"][0] in self.producer else None
                if tr is None or tr["op"] != "Transpose" or _ints(self.m, tr, "perm", None) != [0, 2, 3, 1] or tr["inputs"][0] not in conv_out:
                    raise LowerError("output %s: level tensor %r is not Reshape(Transpose(conv, perm 0 2 3 1))" % (oname, t))
                rec = conv_out[tr["inputs"][0]]
                if rec["act"] != M.ACT_NONE or rec["res"] is not None or rec["w"].shape[0] % k:
                    raise LowerError("output %s: head convolution %s (%d channels) does not end in rows of %d values" % (oname, rec["name"], rec["w"].shape[0], k))
                levels.append((tr["inputs"][0], rec["w"].shape[0] // k))
            found[k] = (oname, levels)
        if 4 not in found or len(found) != 2:
            raise LowerError("outputs %s: one must carry 4 box values per anchor, the other the class logits" % ([o for o, _ in self.m.outputs],))
        reg = found[4]
        cls = found[[k for k in found if k != 4][0]] if len([k for k in found if k != 4]) == 1 else None
        if cls is None or len(reg[1]) != len(cls[1]) or any(a != b for (_, a), (_, b) in zip(reg[1], cls[1])):
            raise LowerError("box and class heads disagree on levels / anchors per cell")
        nc = [k for k in found if k != 4][0]
        return dict(kind="raw", reg=[t for t, _ in reg[1]], cls=[t for t, _ in cls[1]], anchors=reg[1][0][1], nc=nc,
                    f32_tensors=[t for t, _ in reg[1]] + [t for t, _ in cls[1]])

    def _anchors_behind(self, node_idx):
        """The (3, 2) anchor sizes (pixels) of one v5-layout level: the constant of shape (1, 3, ., ., 2) that does not vary over the cells
        (anchor_grid; the other such constant is the cell grid) among the constants consumed downstream of the level's Reshape."""
        seen, todo = set(), [self.nodes[node_idx]["outputs"][0]]
        while todo:
            t = todo.pop()
            for j in self.consumers.get(t, []):
                nd = self.nodes[j]
                if nd["op"] == "Concat" and int(nd["attrs"].get("axis", 0)) == 1 and len(nd["inputs"]) == 3 and nd["outputs"][0] in self.graph_outs:
                    continue
                for x in nd["inputs"]:
                    c = _const(self.m, x)
                    if c is not None and np.asarray(c).ndim == 5 and np.asarray(c).shape[1] == 3 and np.asarray(c).shape[4] == 2:
                        a = np.asarray(c, np.float32)
                        flat = a.reshape(3, -1, 2)
                        if np.all(flat == flat[:, :1]) and float(np.abs(flat).max()) > 0:
                            return flat[:, 0].copy()
                for o in nd["outputs"]:
                    if o not in seen:
                        seen.add(o)
                        todo.append(o)
        raise LowerError("v5-layout Detect level: no anchor_grid constant of shape (1, 3, h, w, 2) found behind the head")

    def _emit_tail(self, tail):
        g = self.g
        nc = tail["nc"]
        if tail["kind"] == "raw":
            for lv, (r, c) in enumerate(zip(tail["reg"], tail["cls"])):
                vr, vc = self._view(r), self._view(c)
                g.output(vr, 0, [1, vr.h * vr.w * tail["anchors"], 4], "regression.l%d" % lv)
                g.output(vc, 0, [1, vc.h * vc.w * tail["anchors"], nc], "classification.l%d" % lv)
            return
        if tail["kind"] == "v8":
            ins, strides = [], []
            for b, c in tail["levels"]:
                vb, vc = self._view(b), self._view(c)
                ins += [vb, vc]
                strides.append(self.in_h // vb.h)
            A = sum(ins[2 * k].h * ins[2 * k].w for k in range(3))
            head = g.buf(1, 1, (4 + nc) * A, f32=True)
            g._op(M.OP_DETECT_V8, ins, head, params=[nc, A] + strides, name="detect.decode")
            g.output(head, 0, [1, 4 + nc, A], tail["out_name"])
            g.meta = dict(kind="yolov8", nc=nc, anchors=A, strides=strides)
        else:
            lv = [(self._view(t), a) for t, a in tail["levels"]]          # graph order = the order of the rows in the output
            ins = [v for v, _ in lv]
            strides = [self.in_h // v.h for v in ins]
            A = 3 * sum(v.h * v.w for v in ins)
            no = nc + 5
            head = g.buf(1, 1, A * no, f32=True)
            anc = np.stack([a for _, a in lv]).astype(np.float32)          # (3 levels, 3 anchors, 2) in pixels
            g._op(M.OP_DETECT_V5, ins, head, w=g._blob(anc), params=[nc, A] + strides, name="detect.decode")
            g.output(head, 0, [1, A, no], tail["out_name"])
            g.meta = dict(kind="yolov5", nc=nc, anchors=A, strides=strides)


def lower(m, name="onnx_graph"):
    """OnnxModel (onnx_import.read_onnx) -> models.Graph.  Raises LowerError (a ValueError) naming the first node it cannot map."""
    return _Lowering(m, name).run()
