"""models.Graph -> ONNX file, in the node vocabulary the exporters of the reference's models emit (ultralytics / torch.onnx.export):
Conv (+ Sigmoid, Mul = SiLU | Relu | LeakyRelu) [+ Add], grouped Conv, MaxPool, AveragePool, Resize(nearest, x2), Slice / Split, Concat,
and the Detect tail (Concat -> Reshape -> Concat -> Split -> DFL softmax-expectation conv -> dist2bbox arithmetic -> Sigmoid -> Concat for the
v8 layout; Reshape -> Transpose -> Sigmoid -> grid / anchor arithmetic for the v5 layout).

TEST INFRASTRUCTURE: lets the suite write ONNX files of graphs that models.py does NOT hand-build (other widths / depths, ad-hoc
topologies) and check that vehicle-cv-adas_amd/onnx_lower.py maps them back onto the engine's op list.  Shares no code with the lowering."""
import importlib

import numpy as np

import onnx_writer as OW
from conftest import load_pkg

load_pkg()
M = importlib.import_module("adas_amd.models")


def _i64(name, vals):
    return OW.tensor(name, np.asarray(vals, np.int64))


class Emitter:
    def __init__(self, g, use_split=True, hswish_as_mul=False):
        self.g = g
        self.hswish_as_mul = hswish_as_mul     # hard-swish as x * HardSigmoid(x) (exports below opset 14) instead of a HardSwish node
        self.blob = np.frombuffer(bytes(g.blob), np.float32)
        self.nodes, self.inits = [], []
        self.segs = {}          # buffer -> list of (coff, c, tensor name), latest writer wins
        self.n = 0
        self.use_split = use_split
        self.split_cache = {}

    def name(self, base):
        self.n += 1
        return "%s_%d" % (base, self.n)

    def node(self, op, ins, outs, attrs=(), name=None):
        self.nodes.append(OW.node(op, ins, outs, name or self.name("/" + op), attrs))

    def wrote(self, view, tensor):
        """Segments are (buffer channel offset, channels, tensor, offset inside the tensor, the tensor's channel count): a later write over PART
        of an earlier tensor's range (PSA: b + ffn(b) lands in b's slot of the (a, b) buffer) keeps the rest of that tensor readable."""
        lo, hi = view.coff, view.coff + view.c
        segs = []
        for off, c, t, toff, tc in self.segs.get(view.buf, []):
            if off + c <= lo or off >= hi:
                segs.append((off, c, t, toff, tc))
                continue
            if off < lo:
                segs.append((off, lo - off, t, toff, tc))
            if off + c > hi:
                segs.append((hi, off + c - hi, t, toff + (hi - off), tc))
        segs.append((view.coff, view.c, tensor, 0, view.c))
        self.segs[view.buf] = sorted(segs)

    def slice_of(self, tensor, total_c, off, c):
        if off == 0 and c == total_c:
            return tensor
        if self.use_split and total_c == 2 * c and off in (0, c):          # x.chunk(2, 1): one Split node, both halves
            key = (tensor, c)
            if key not in self.split_cache:
                a, b = self.name(tensor + "_s0"), self.name(tensor + "_s1")
                sp = self.name("split")
                self.inits.append(_i64(sp, [c, c]))
                self.node("Split", [tensor, sp], [a, b], [OW.attr_int("axis", 1)])
                self.split_cache[key] = (a, b)
            return self.split_cache[key][0 if off == 0 else 1]
        if (tensor, off, c) in self.split_cache:
            return self.split_cache[(tensor, off, c)]
        out = self.name(tensor + "_sl")
        self.split_cache[(tensor, off, c)] = out
        st, en, ax = self.name("starts"), self.name("ends"), self.name("axes")
        self.inits += [_i64(st, [off]), _i64(en, [off + c]), _i64(ax, [1])]
        self.node("Slice", [tensor, st, en, ax], [out])
        return out

    def read(self, view):
        """ONNX tensor holding the channels [coff, coff + c) of the view's buffer: a produced tensor, a slice of one, or a Concat."""
        parts, pos, end = [], view.coff, view.coff + view.c
        for off, c, t, toff, tc in self.segs.get(view.buf, []):
            if off + c <= pos or off >= end:
                continue
            assert off <= pos, ("hole in buffer %d at channel %d" % (view.buf, pos))
            lo, hi = pos - off, min(end, off + c) - off
            parts.append(self.slice_of(t, tc, toff + lo, hi - lo))
            pos = off + hi
        assert pos == end and parts, (view.buf, view.coff, view.c, self.segs.get(view.buf))
        if len(parts) == 1:
            return parts[0]
        out = self.name("cat")
        self.node("Concat", parts, [out], [OW.attr_int("axis", 1)])
        return out

    def act(self, x, a):
        if a == M.ACT_SILU:
            s, o = self.name("sig"), self.name("silu")
            self.node("Sigmoid", [x], [s])
            self.node("Mul", [x, s], [o])
            return o
        if a == M.ACT_RELU:
            o = self.name("relu")
            self.node("Relu", [x], [o])
            return o
        if a == M.ACT_LEAKY:
            o = self.name("lrelu")
            self.node("LeakyRelu", [x], [o], [OW.attr_float("alpha", 0.1)])
            return o
        return x

    def emit(self, path, out_name="output0"):
        g = self.g
        first_conv = True
        outputs = []
        for op in g.ops:
            t, ins, out = op["type"], op["ins"], op["out"]
            (wo, wn), (bo, bn) = op["w"], op["b"]
            w, b = self.blob[wo // 4: wo // 4 + wn], self.blob[bo // 4: bo // 4 + bn]
            if t == M.OP_INPUT:
                self.wrote(M.View(out.buf, 0, 8, out.h, out.w), "images")
            elif t == M.OP_CONV:
                k, cout, cin_buf = op["kh"], out.c, ins[0].c
                if first_conv:
                    cin = g.in_c
                    W = w.reshape(cout, k, k, cin_buf)[..., :cin].transpose(0, 3, 1, 2)
                    x = "images"
                    first_conv = False
                else:
                    cin = cin_buf
                    W = w.reshape(cout, cin, 1, 1) if (k == 1 and w.size == cout * cin) else w.reshape(cout, k, k, cin).transpose(0, 3, 1, 2)
                    x = self.read(ins[0])
                base = op["name"]
                self.inits += [OW.tensor(base + ".weight", np.ascontiguousarray(W, np.float32)), OW.tensor(base + ".bias", np.ascontiguousarray(b, np.float32))]
                y = self.name(base + "_out")
                p = op["pad"]
                self.node("Conv", [x, base + ".weight", base + ".bias"], [y],
                          [OW.attr_ints("kernel_shape", [k, k]), OW.attr_ints("strides", [op["stride"]] * 2), OW.attr_ints("pads", [p] * 4),
                           OW.attr_ints("dilations", [1, 1]), OW.attr_int("group", 1)], name="/" + base + "/Conv")
                if op["res_mode"] == M.RES_BEFORE_ACT:
                    s_ = self.name("add")
                    self.node("Add", [y, self.read(op["res"])], [s_])
                    y = self.act(s_, op["act"])
                else:
                    y = self.act(y, op["act"])
                    if op["res_mode"] == M.RES_AFTER_ACT:
                        s_ = self.name("add")
                        self.node("Add", [self.read(op["res"]), y], [s_])
                        y = s_
                self.wrote(out, y)
            elif t == M.OP_ATTENTION:
                # ultralytics Attention.forward as torch.onnx exports it: qkv -> Reshape (B, heads, 2 kd + hd, N) -> Split -> q^T k * scale ->
                # Softmax -> v attn^T -> Reshape (B, C, H, W), plus pe = depth-wise conv of v.reshape(B, C, H, W), Add.  The engine graph holds
                # the pe conv as one depth-wise op per head (each adds its slice of the attention output): they are gathered here and
                # written as the ONE grouped convolution the exporter emits.
                nh, kd, hd, scale = int(op["params"][0]), int(op["params"][1]), int(op["params"][2]), float(op["params"][3])
                H_, W_ = ins[0].h, ins[0].w
                N_, C_ = H_ * W_, nh * hd
                qkv = self.read(ins[0])
                heads = [o2 for o2 in g.ops if o2["type"] == M.OP_DWCONV and o2["ins"][0].buf == ins[0].buf and o2["res"] is not None and o2["res"].buf == out.buf]
                assert len(heads) == nh and all(o2["act"] == M.ACT_NONE and o2["kh"] == heads[0]["kh"] for o2 in heads), "attention without its per-head pe convs"
                heads.sort(key=lambda o2: o2["res"].coff)
                names = {n: self.name(n) for n in ("r", "q", "k", "v", "qt", "s", "sc", "a", "at", "o", "o4", "v4", "pe", "y", "shp", "shp4", "spl", "scale")}
                self.inits += [_i64(names["shp"], [1, nh, 2 * kd + hd, N_]), _i64(names["shp4"], [1, C_, H_, W_]), _i64(names["spl"], [kd, kd, hd]),
                               OW.tensor(names["scale"], np.asarray(np.float32(scale)).reshape(()))]
                self.node("Reshape", [qkv, names["shp"]], [names["r"]])
                self.node("Split", [names["r"], names["spl"]], [names["q"], names["k"], names["v"]], [OW.attr_int("axis", 2)])
                self.node("Transpose", [names["q"]], [names["qt"]], [OW.attr_ints("perm", [0, 1, 3, 2])])
                self.node("MatMul", [names["qt"], names["k"]], [names["s"]])
                self.node("Mul", [names["s"], names["scale"]], [names["sc"]])
                self.node("Softmax", [names["sc"]], [names["a"]], [OW.attr_int("axis", -1)])
                self.node("Transpose", [names["a"]], [names["at"]], [OW.attr_ints("perm", [0, 1, 3, 2])])
                self.node("MatMul", [names["v"], names["at"]], [names["o"]])
                self.node("Reshape", [names["o"], names["shp4"]], [names["o4"]])
                self.node("Reshape", [names["v"], names["shp4"]], [names["v4"]])
                kpe = heads[0]["kh"]
                pw = np.concatenate([self.blob[o2["w"][0] // 4: o2["w"][0] // 4 + o2["w"][1]].reshape(hd, 1, kpe, kpe) for o2 in heads])
                pb = np.concatenate([self.blob[o2["b"][0] // 4: o2["b"][0] // 4 + o2["b"][1]] for o2 in heads])
                base = heads[0]["name"].rsplit(".h", 1)[0]
                self.inits += [OW.tensor(base + ".weight", np.ascontiguousarray(pw, np.float32)), OW.tensor(base + ".bias", np.ascontiguousarray(pb, np.float32))]
                self.node("Conv", [names["v4"], base + ".weight", base + ".bias"], [names["pe"]],
                          [OW.attr_ints("kernel_shape", [kpe, kpe]), OW.attr_ints("pads", [kpe // 2] * 4), OW.attr_int("group", C_)])
                self.node("Add", [names["o4"], names["pe"]], [names["y"]])
                summed = heads[0]["out"]
                self.wrote(M.View(summed.buf, summed.coff - heads[0]["res"].coff, C_, H_, W_), names["y"])
                self.pe_done = getattr(self, "pe_done", set()) | {id(o2) for o2 in heads}
            elif t == M.OP_DWCONV and id(op) in getattr(self, "pe_done", set()):
                pass      # written with its attention op
            elif t == M.OP_DWCONV:
                k, c = op["kh"], out.c
                base = op["name"]
                self.inits += [OW.tensor(base + ".weight", np.ascontiguousarray(w.reshape(c, 1, k, k), np.float32)), OW.tensor(base + ".bias", np.ascontiguousarray(b, np.float32))]
                y = self.name(base + "_out")
                self.node("Conv", [self.read(ins[0]), base + ".weight", base + ".bias"], [y],
                          [OW.attr_ints("kernel_shape", [k, k]), OW.attr_ints("strides", [op["stride"]] * 2), OW.attr_ints("pads", [op["pad"]] * 4),
                           OW.attr_ints("dilations", [1, 1]), OW.attr_int("group", c)])
                y = self.act(y, op["act"])
                if op["res_mode"] != M.RES_NONE:
                    s_ = self.name("add")
                    self.node("Add", [self.read(op["res"]), y], [s_])
                    y = s_
                self.wrote(out, y)
            elif t in (M.OP_MAXPOOL, M.OP_AVGPOOL):
                y = self.name("pool")
                attrs = [OW.attr_ints("kernel_shape", [op["kh"]] * 2), OW.attr_ints("strides", [op["stride"]] * 2), OW.attr_ints("pads", [op["pad"]] * 4)]
                if t == M.OP_AVGPOOL:
                    attrs.append(OW.attr_int("count_include_pad", 1))
                self.node("MaxPool" if t == M.OP_MAXPOOL else "AveragePool", [self.read(ins[0])], [y], attrs)
                self.wrote(out, y)
            elif t == M.OP_SHUFFLE:
                gr, c_ = int(op["params"][0]), ins[0].c
                s1, s2, r1, t1, y = self.name("shape"), self.name("shape"), self.name("sh_r"), self.name("sh_t"), self.name("shuffled")
                self.inits += [_i64(s1, [1, gr, c_ // gr, ins[0].h, ins[0].w]), _i64(s2, [1, c_, ins[0].h, ins[0].w])]
                self.node("Reshape", [self.read(ins[0]), s1], [r1])
                self.node("Transpose", [r1], [t1], [OW.attr_ints("perm", [0, 2, 1, 3, 4])])
                self.node("Reshape", [t1, s2], [y])
                self.wrote(out, y)
            elif t == M.OP_SE_GATE:
                # squeeze-and-excitation as torch exports it: GlobalAveragePool -> Conv -> Sigmoid * Mul (swish) -> Conv -> Sigmoid;  the
                # gate tensor (N, C, 1, 1) multiplies x in the OP_SCALE that follows
                c, cr = ins[0].c, int(op["params"][0])
                w_, b_ = self.blob[wo // 4: wo // 4 + wn], self.blob[bo // 4: bo // 4 + bn]
                base = op["name"][:-len(".gate")] if op["name"].endswith(".gate") else op["name"]
                n1, n2 = base + ".reduce", base + ".expand"
                self.inits += [OW.tensor(n1 + ".weight", w_[:cr * c].reshape(cr, c, 1, 1)), OW.tensor(n1 + ".bias", w_[cr * c:]),
                               OW.tensor(n2 + ".weight", b_[:c * cr].reshape(c, cr, 1, 1)), OW.tensor(n2 + ".bias", b_[c * cr:])]
                gp, r1, sg, a1, e1, gate = (self.name(n) for n in ("gap", "se_r", "se_sig", "se_a", "se_e", "se_gate"))
                self.node("GlobalAveragePool", [self.read(ins[0])], [gp])
                prm = list(op["params"]) + [0, 0, 0]
                self.node("Conv", [gp, n1 + ".weight", n1 + ".bias"], [r1], [OW.attr_ints("kernel_shape", [1, 1])])
                if int(prm[1]) == M.ACT_RELU:             # MobileNetV3 / PP-LCNet form: ReLU inside, hard-sigmoid gate
                    self.node("Relu", [r1], [a1])
                else:
                    self.node("Sigmoid", [r1], [sg])
                    self.node("Mul", [r1, sg], [a1])
                self.node("Conv", [a1, n2 + ".weight", n2 + ".bias"], [e1], [OW.attr_ints("kernel_shape", [1, 1])])
                if int(prm[2]) == M.ACT_HSIGMOID:
                    self.node("HardSigmoid", [e1], [gate], [OW.attr_float("alpha", 1.0 / 6.0), OW.attr_float("beta", 0.5)])
                else:
                    self.node("Sigmoid", [e1], [gate])
                self.wrote(out, gate)
            elif t == M.OP_SCALE:
                y = self.name("se_out")
                self.node("Mul", [self.read(ins[0]), self.read(ins[1])], [y])
                self.wrote(out, y)
            elif t == M.OP_WSUM:
                # BiFPN / CBFuse style node as exporters write it: [Resize(nearest, x2)] -> Mul(constant) per input, a chain of Adds, activation
                terms = []
                for v, wgt in zip(ins, op["params"]):
                    a = self.read(v)
                    if v.h * 2 == out.h:
                        u, sc = self.name("up"), self.name("scales")
                        self.inits.append(OW.tensor(sc, np.asarray([1, 1, 2, 2], np.float32)))
                        self.node("Resize", [a, "", sc], [u], [OW.attr_str("mode", "nearest")])
                        a = u
                    if float(np.float32(wgt)) != 1.0:
                        cw, mu = self.name("fw"), self.name("scaled")
                        self.inits.append(OW.tensor(cw, np.asarray(np.float32(wgt)).reshape(())))
                        self.node("Mul", [a, cw], [mu])
                        a = mu
                    terms.append(a)
                acc = terms[0]
                for a in terms[1:]:
                    nx = self.name("sum")
                    self.node("Add", [acc, a], [nx])
                    acc = nx
                if op["act"] == M.ACT_SILU:
                    sg, y = self.name("sig"), self.name("act")
                    self.node("Sigmoid", [acc], [sg])
                    self.node("Mul", [acc, sg], [y])
                    acc = y
                elif op["act"] == M.ACT_RELU:
                    y = self.name("act")
                    self.node("Relu", [acc], [y])
                    acc = y
                elif op["act"] == M.ACT_LEAKY:
                    y = self.name("act")
                    self.node("LeakyRelu", [acc], [y], [OW.attr_float("alpha", 0.1)])
                    acc = y
                elif op["act"] == M.ACT_HSWISH:
                    y = self.name("act")
                    if self.hswish_as_mul:        # opset < 14 exporters: x * HardSigmoid(x)
                        hs = self.name("hsig")
                        self.node("HardSigmoid", [acc], [hs], [OW.attr_float("alpha", 1.0 / 6.0), OW.attr_float("beta", 0.5)])
                        self.node("Mul", [acc, hs], [y])
                    else:
                        self.node("HardSwish", [acc], [y])
                    acc = y
                elif op["act"] == M.ACT_HSIGMOID:
                    y = self.name("act")
                    self.node("HardSigmoid", [acc], [y], [OW.attr_float("alpha", 1.0 / 6.0), OW.attr_float("beta", 0.5)])
                    acc = y
                elif op["act"] == M.ACT_RELU6:          # torch.nn.ReLU6 exports as Clip(x, 0, 6): bounds as inputs (opset >= 11) or attributes
                    y = self.name("act")
                    if self.hswish_as_mul:              # (the "older export" switch)
                        self.node("Clip", [acc], [y], [OW.attr_float("min", 0.0), OW.attr_float("max", 6.0)])
                    else:
                        lo, hi = self.name("clip_lo"), self.name("clip_hi")
                        self.inits += [OW.tensor(lo, np.asarray(0.0, np.float32).reshape(())), OW.tensor(hi, np.asarray(6.0, np.float32).reshape(()))]
                        self.node("Clip", [acc, lo, hi], [y])
                    acc = y
                else:
                    assert op["act"] == M.ACT_NONE, op["act"]
                self.wrote(out, acc)
            elif t == M.OP_UPSAMPLE2:
                y, sc = self.name("up"), self.name("scales")
                self.inits.append(OW.tensor(sc, np.asarray([1, 1, 2, 2], np.float32)))
                self.node("Resize", [self.read(ins[0]), "", sc], [y], [OW.attr_str("mode", "nearest")])
                self.wrote(out, y)
            elif t == M.OP_DETECT_V8:
                nc, A = int(op["params"][0]), int(op["params"][1])
                lv = []
                for i in range(3):
                    c_, r_ = self.name("dcat"), self.name("dresh")
                    self.node("Concat", [self.read(ins[2 * i]), self.read(ins[2 * i + 1])], [c_], [OW.attr_int("axis", 1)])
                    shp = self.name("shape")
                    self.inits.append(_i64(shp, [1, 64 + nc, -1]))
                    self.node("Reshape", [c_, shp], [r_])
                    lv.append(r_)
                allc = self.name("dall")
                self.node("Concat", lv, [allc], [OW.attr_int("axis", 2)])
                # the rest of the tail (DFL expectation, dist2bbox, sigmoid) as the exporter writes it; the lowering recognises the head
                # by its front (three Concat -> Reshape pairs feeding one axis-2 Concat) and replaces everything behind it
                bx, cl, sp = self.name("box"), self.name("cls"), self.name("split")
                self.inits.append(_i64(sp, [64, nc]))
                self.node("Split", [allc, sp], [bx, cl], [OW.attr_int("axis", 1)])
                s1, r1, t1, sm, dfl, r2 = (self.name(n) for n in ("shape", "r", "t", "sm", "dfl", "r"))
                self.inits += [_i64(s1, [1, 4, 16, A]), OW.tensor("dfl.conv.weight", np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1))]
                self.node("Reshape", [bx, s1], [r1])
                self.node("Transpose", [r1], [t1], [OW.attr_ints("perm", [0, 2, 1, 3])])
                self.node("Softmax", [t1], [sm], [OW.attr_int("axis", 1)])
                self.node("Conv", [sm, "dfl.conv.weight"], [dfl], [OW.attr_ints("kernel_shape", [1, 1])])
                s2 = self.name("shape")
                self.inits.append(_i64(s2, [1, 4, A]))
                self.node("Reshape", [dfl, s2], [r2])
                sg = self.name("clsp")
                self.node("Sigmoid", [cl], [sg])
                self.node("Concat", [r2, sg], [out_name], [OW.attr_int("axis", 1)])
                outputs.append((out_name, [1, 4 + nc, A]))
            elif t == M.OP_DETECT_V5:
                nc, A = int(op["params"][0]), int(op["params"][1])
                no = nc + 5
                anc = self.blob[wo // 4: wo // 4 + wn].reshape(3, 3, 2)
                rows = []
                for l in range(3):
                    h, w_ = ins[l].h, ins[l].w
                    s5, r5, t5, sg = (self.name(n) for n in ("shape", "r5", "t5", "sig"))
                    self.inits.append(_i64(s5, [1, 3, no, h, w_]))
                    self.node("Reshape", [self.read(ins[l]), s5], [r5])
                    self.node("Transpose", [r5], [t5], [OW.attr_ints("perm", [0, 1, 3, 4, 2])])
                    self.node("Sigmoid", [t5], [sg])
                    # wh branch: (2 p)^2 * anchor_grid -- the constant the lowering reads the anchors from
                    ag, two, m2, pw, wh = (self.name(n) for n in ("anchor_grid", "two", "m2", "pow", "wh"))
                    self.inits += [OW.tensor(ag, np.ascontiguousarray(np.broadcast_to(anc[l].reshape(1, 3, 1, 1, 2), (1, 3, h, w_, 2)), np.float32)),
                                   OW.tensor(two, np.asarray([2.0], np.float32))]
                    self.node("Mul", [sg, two], [m2])
                    self.node("Pow", [m2, two], [pw])
                    self.node("Mul", [pw, ag], [wh])
                    s3, r3 = self.name("shape"), self.name("rows")
                    self.inits.append(_i64(s3, [1, -1, no]))
                    self.node("Reshape", [wh, s3], [r3])
                    rows.append(r3)
                self.node("Concat", rows, [out_name], [OW.attr_int("axis", 1)])
                outputs.append((out_name, [1, A, no]))
            else:
                raise NotImplementedError("emitter: op type %d (%s)" % (t, op["name"]))
        if not outputs and g.outs and all(nm.startswith(("regression.l", "classification.l")) for _, _, _, nm in g.outs):
            # head-only export of an anchor-based detector (EfficientDet without decode / NMS): per level conv -> Transpose(NHWC) -> Reshape
            # (1, -1, k), levels concatenated along axis 1: two outputs
            cols = {"regression": [], "classification": []}
            per = {}
            for buf, off, dims, nm in g.outs:
                kind = nm.split(".")[0]
                h_, w_, c_, _ = g.bufs[buf]
                t = self.read(M.View(buf, 0, c_, h_, w_))
                tr, rs, shp = self.name("nhwc"), self.name("rows"), self.name("shape")
                self.node("Transpose", [t], [tr], [OW.attr_ints("perm", [0, 2, 3, 1])])
                self.inits.append(_i64(shp, [1, -1, int(dims[2])]))
                self.node("Reshape", [tr, shp], [rs])
                cols[kind].append(rs)
                per[kind] = (int(dims[2]), per.get(kind, (0, 0))[1] + int(dims[1]))
            for kind in ("regression", "classification"):
                self.node("Concat", cols[kind], [kind], [OW.attr_int("axis", 1)])
                outputs.append((kind, [1, per[kind][1], per[kind][0]]))
        assert outputs, "graph has no Detect op (the emitter writes detector graphs)"
        data = OW.model(self.nodes, self.inits, [("images", [1, g.in_c, g.in_h, g.in_w])], outputs)
        open(path, "wb").write(data)
        return path


def emit(g, path, use_split=True, hswish_as_mul=False):
    return Emitter(g, use_split, hswish_as_mul).emit(path)
