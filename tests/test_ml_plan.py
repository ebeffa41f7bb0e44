"""CPU: the planner of the multi-layer persistent launch (csrc/conv_ml.hip, include/adas_hip.h adas_debug_ml_plan) on layer descriptions
alone -- no device.  The kernel's progress argument (every wait is for items with SMALLER tickets) and its correctness argument (an
item waits for every layer that writes what it reads, or reads / writes what it overwrites) are properties of the tables, checked here
against an independent restatement: hazards from the views, frame-complete closure of the waited-for layers, ticket order."""
import ctypes as C
import importlib
import itertools
import os

import numpy as np
import pytest

from conftest import load_pkg

load_pkg()
L = importlib.import_module("adas_amd._lib")
M = importlib.import_module("adas_amd.models")

HALO, PW = 1, 4          # CONV_HALO, CONV_PW (csrc/kernels.h)
SILU, NONE_ = M.ACT_SILU, M.ACT_NONE


class Net:
    """A layer list in launch order over named buffers (views = channel slices), as the engine hands it to ml_plan_create."""

    def __init__(self):
        self.layers, self.bufs = [], {}

    def buf(self, name, h, w, c):
        self.bufs[name] = (0x100000 * (len(self.bufs) + 1), h, w, c)
        return name

    def view(self, name, coff=0, c=None):
        base, h, w, cs = self.bufs[name]
        return L.MlView(base, cs, coff, cs - coff if c is None else c, h, w)

    def conv3(self, x, y, stride=1, res=None, act=SILU):
        d = L.MlLayerDesc(HALO, stride, act, M.RES_AFTER_ACT if res is not None else M.RES_NONE, 0, 0, x, y, res if res is not None else L.MlView(), L.MlView())
        self.layers.append(d)

    def conv1(self, x, y, up=None, act=SILU):
        d = L.MlLayerDesc(PW, 1, act, M.RES_NONE, up.c if up is not None else 0, 0, x, y, L.MlView(), up if up is not None else L.MlView())
        self.layers.append(d)


def reads_of(d):
    r = []
    if d.kernel == PW and d.up_c > 0:
        r.append((d.up.buf, d.up.coff, d.up.coff + d.up.c))
        r.append((d.x.buf, d.x.coff + d.up_c, d.x.coff + d.x.c))
    else:
        r.append((d.x.buf, d.x.coff, d.x.coff + d.x.c))
    if d.res_mode != M.RES_NONE:
        r.append((d.res.buf, d.res.coff, d.res.coff + d.y.c))
    return r


def write_of(d):
    return (d.y.buf, d.y.coff, d.y.coff + d.y.c)


def ov(a, b):
    return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]


def plan(net, batch, order=None):
    n = len(net.layers)
    arr = (L.MlLayerDesc * n)(*net.layers)
    deps, tg, summ = (C.c_int32 * (6 * n))(), (C.c_int32 * (6 * n))(), (C.c_int32 * 4)()
    old = os.environ.get("ADAS_ML_ORDER")
    if order is not None:
        os.environ["ADAS_ML_ORDER"] = str(order)
    try:
        L.check(L.lib().adas_debug_ml_plan(arr, n, batch, L.PREC_FP16, deps, tg, None, 0, summ))
        items = (C.c_uint64 * summ[0])()
        L.check(L.lib().adas_debug_ml_plan(arr, n, batch, L.PREC_FP16, deps, tg, items, summ[0], summ))
    finally:
        if order is not None:
            if old is None:
                del os.environ["ADAS_ML_ORDER"]
            else:
                os.environ["ADAS_ML_ORDER"] = old
    deps = [[deps[i * 6 + k] for k in range(6) if deps[i * 6 + k] >= 0] for i in range(n)]
    tg = [[tg[i * 6 + k] for k in range(len(deps[i]))] for i in range(n)]
    w = np.frombuffer(items, np.uint64)
    hi = (w >> np.uint64(32)).astype(np.int64)
    tab = np.stack([(hi & 255), (hi >> 8) & 255, hi >> 16, (w & np.uint64(0xffffffff)).astype(np.int64)], 1)   # layer, cb, frame, tile
    return deps, tg, tab, list(summ)


def check_plan(net, batch, deps, tg, tab):
    n = len(net.layers)
    # (1) every hazard is covered by the frame-complete closure of the layers waited for
    closure = []
    for i in range(n):
        c = {i}
        for d in deps[i]:
            assert d < i
            c |= closure[d]
        closure.append(c)
    hazards = 0
    for i in range(n):
        for j in range(i):
            raw = any(ov(r, write_of(net.layers[j])) for r in reads_of(net.layers[i]))
            war = any(ov(r, write_of(net.layers[i])) for r in reads_of(net.layers[j]))
            waw = ov(write_of(net.layers[i]), write_of(net.layers[j]))
            if raw or war or waw:
                hazards += 1
                assert j in closure[i] - {i}, (i, j, deps[i])
        # reduction: nothing waited for twice over (no dep implied by another dep)
        for a, b in itertools.permutations(deps[i], 2):
            assert a not in closure[b], (i, a, b)
    # (2) the item table: per (layer, frame) the same count for every frame = the target its consumers wait for; complete; frames in range
    assert tab[:, 0].max() == n - 1 and tab[:, 2].max() == batch - 1
    per = np.zeros((n, batch), np.int64)
    np.add.at(per, (tab[:, 0], tab[:, 2]), 1)
    assert (per == per[:, :1]).all() and (per > 0).all()
    for i in range(n):
        for d, t in zip(deps[i], tg[i]):
            assert t == per[d, 0], (i, d, t, per[d, 0])
    # no duplicated item
    assert len({tuple(r) for r in tab.tolist()}) == len(tab)
    # (3) ticket order: when an item is handed out, all items of its producers' groups (same frame) hold smaller tickets
    seen = np.zeros((n, batch), np.int64)
    for layer, cb, frame, tile in tab.tolist():
        for d in deps[layer]:
            assert seen[d, frame] == per[d, 0], (layer, frame, d)
        seen[layer, frame] += 1
    return hazards


def yolov8n_tail(batch_hw=(40, 20)):
    """The neck / head layers of YOLOv8n behind model.15 as the engine offers them at 64 frames (segment shapes of models.yolov8("n")):
    stride-2 conv into a concat buffer, C2f blocks (cv1 -> split -> 3x3 -> 3x3 -> cv2 over the concat), the next level, Detect branches."""
    H4, H5 = batch_hw
    H3 = 2 * H4
    g = Net()
    g.buf("p3", H3, H3, 64); g.buf("cat17", H4, H4, 192); g.buf("c18", H4, H4, 192); g.buf("t18", H4, H4, 64); g.buf("p4", H4, H4, 128)
    g.buf("cat20", H5, H5, 384); g.buf("c21", H5, H5, 384); g.buf("t21", H5, H5, 128); g.buf("p5", H5, H5, 256)
    # cat17 = [model.16 out (64) | n12 (128, written before the launch)]
    g.conv3(g.view("p3"), g.view("cat17", 0, 64), stride=2)                               # 0 model.16
    g.conv1(g.view("cat17"), g.view("c18", 0, 128))                                      # 1 model.18.cv1
    g.conv3(g.view("c18", 64, 64), g.view("t18"))                                        # 2 m.0.cv1
    g.conv3(g.view("t18"), g.view("c18", 128, 64))                                       # 3 m.0.cv2
    g.conv1(g.view("c18"), g.view("p4"))                                                 # 4 model.18.cv2
    g.conv3(g.view("p4"), g.view("cat20", 0, 128), stride=2)                             # 5 model.19
    g.conv1(g.view("cat20"), g.view("c21", 0, 256))                                      # 6 model.21.cv1
    g.conv3(g.view("c21", 128, 128), g.view("t21"))                                      # 7
    g.conv3(g.view("t21"), g.view("c21", 256, 128))                                      # 8
    g.conv1(g.view("c21"), g.view("p5"))                                                 # 9 model.21.cv2
    # Detect: cv3.0.* on P3 (the 64 -> 64 box branch of P3 runs on conv_halo_rw: not offered), both branches on P4 and P5
    for name, c in (("d3c0", 80), ("d3c1", 80), ("d4b0", 64), ("d4b1", 64), ("d4c0", 80), ("d4c1", 80), ("d5b0", 64), ("d5b1", 64), ("d5c0", 80), ("d5c1", 80)):
        g.buf(name, {"3": H3, "4": H4, "5": H5}[name[1]], {"3": H3, "4": H4, "5": H5}[name[1]], c)
    g.conv3(g.view("p3"), g.view("d3c0")); g.conv3(g.view("d3c0"), g.view("d3c1"))       # 10, 11
    g.conv3(g.view("p4"), g.view("d4b0")); g.conv3(g.view("d4b0"), g.view("d4b1"))       # 12, 13
    g.conv3(g.view("p4"), g.view("d4c0")); g.conv3(g.view("d4c0"), g.view("d4c1"))       # 14, 15
    g.conv3(g.view("p5"), g.view("d5b0")); g.conv3(g.view("d5b0"), g.view("d5b1"))       # 16, 17
    g.conv3(g.view("p5"), g.view("d5c0")); g.conv3(g.view("d5c0"), g.view("d5c1"))       # 18, 19
    return g


@pytest.mark.parametrize("order", [0, 1])
def test_yolov8n_tail_plan(order):
    g = yolov8n_tail()
    deps, tg, tab, summ = plan(g, 64, order)
    hz = check_plan(g, 64, deps, tg, tab)
    assert hz >= 19 and summ[0] == len(tab) and summ[3] == order and summ[2] <= 80 * 1024
    # the chain and the branches come out as the graph has them
    assert deps[1] == [0] and deps[2] == [1] and deps[3] == [2] and deps[4] == [3] and deps[5] == [4] and deps[9] == [8]
    assert deps[10] == [] and deps[11] == [10] and deps[12] == [4] and deps[14] == [4] and deps[16] == [9] and deps[19] == [18]
    if order == 1:
        # the list schedule interleaves: the independent P3 Detect branch (layers 10, 11: the launch's largest) does not wait for the
        # chain's end -- its first item is handed out before the LAST item of the 40x40 C2f (layer 4)
        first = {l: int(np.argmax(tab[:, 0] == l)) for l in range(20)}
        last = {l: len(tab) - 1 - int(np.argmax(tab[::-1, 0] == l)) for l in range(20)}
        assert first[10] < last[4] and first[12] < last[9]
    else:
        assert (np.diff(tab[:, 0]) >= 0).all()          # layer-major


def test_c2f_with_shortcut_and_upsample_fold():
    """A C2f with a shortcut Bottleneck (the residual is another slice of the concat buffer the conv writes into) behind a 1x1 conv whose
    leading channels come from a half-resolution tensor produced INSIDE the launch (YOLO neck: model.9.cv2 -> Upsample -> model.12.cv1)."""
    g = Net()
    g.buf("x9", 20, 20, 512); g.buf("cat11", 40, 40, 384); g.buf("p5b", 20, 20, 256); g.buf("c12", 40, 40, 192); g.buf("t", 40, 40, 64); g.buf("o", 40, 40, 128)
    g.conv1(g.view("x9"), g.view("p5b"))                                                           # 0: SPPF cv2 (20x20)
    g.conv1(g.view("cat11"), g.view("c12", 0, 128), up=g.view("p5b"))                               # 1: reads up(p5b) + cat11[256:384]
    g.conv3(g.view("c12", 64, 64), g.view("t"))                                                    # 2
    g.conv3(g.view("t"), g.view("c12", 128, 64), res=g.view("c12", 64, 64))                        # 3: shortcut = the slice conv 2 read
    g.conv1(g.view("c12"), g.view("o"))                                                            # 4
    deps, tg, tab, summ = plan(g, 16)
    check_plan(g, 16, deps, tg, tab)
    assert deps == [[], [0], [1], [2], [3]]
    # the pointwise layer behind the half-resolution producer waits for ALL of that frame's 20x20 items
    assert tg[1] == [int(((tab[:, 0] == 0) & (tab[:, 2] == 0)).sum())]


def test_buffer_reuse_inside_a_launch_is_ordered():
    """Write-after-read: a layer that overwrites a buffer an earlier layer of the launch still reads must wait for that reader."""
    g = Net()
    g.buf("a", 40, 40, 64); g.buf("b", 40, 40, 64); g.buf("c", 40, 40, 64)
    g.conv3(g.view("a"), g.view("b"))        # 0 reads a
    g.conv3(g.view("a"), g.view("c"))        # 1 reads a
    g.conv3(g.view("c"), g.view("a"))        # 2 overwrites a: after 0 and 1
    deps, tg, tab, _ = plan(g, 8)
    check_plan(g, 8, deps, tg, tab)
    assert deps[2] == [0, 1]


def test_unsupported_shapes_are_refused():
    g = Net()
    g.buf("a", 40, 40, 24); g.buf("b", 40, 40, 64)
    g.conv1(g.view("a"), g.view("b"))        # 24 input channels: not a whole K step
    arr = (L.MlLayerDesc * 1)(*g.layers)
    deps, tg, summ = (C.c_int32 * 6)(), (C.c_int32 * 6)(), (C.c_int32 * 4)()
    rc = L.lib().adas_debug_ml_plan(arr, 1, 4, L.PREC_FP16, deps, tg, None, 0, summ)
    assert rc != 0 and b"not supported" in L.lib().adas_last_error()
