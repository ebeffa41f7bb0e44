#!/usr/bin/env python3
"""Pin the lane-network oracle to the REFERENCE's own network modules.

Run in the build container only (needs /root/reference and torch-CPU):
    python tests/golden/make_golden_ufldnet.py
Writes tests/golden/ufld_net.npz.

The reference vendors the networks it exports to ONNX:
  * TrafficLaneDetector/ufldDetector/exportLib/ultrafastLaneV2/model_culane.py:7-63  (UFLDv2 parsingNet; model_tusimple.py
    re-exports the same class with fc_norm=False) on backbone.py:14-58 (`resnet` = torchvision ResNet trunk);
  * TrafficLaneDetector/ufldDetector/exportLib/ultrafastLane/model.py:19-89         (UFLD v1 parsingNet).
Both import `torchvision`, which this image lacks.  The stub below provides `torchvision.models.resnet18/34` as a plain-torch
BasicBlock ResNet with torchvision's module names and forward order (conv1-bn1-relu-maxpool-layer1..4; block = conv3x3-bn-relu-
conv3x3-bn, + identity or conv1x1/bn shortcut, relu) -- that topology is public torchvision API, not reference code.  The
reference's parsingNet classes are imported from where they lie, UNMODIFIED, built with `pretrained=False`, loaded with the seeded
state_dict of tests/golden/ufldnet_params.py (BatchNorm NOT folded: eval-mode running statistics) and run on one seeded frame.
The fixture keeps, per case, an evenly strided sample and the sum / abs-sum of every output (+ the pooled feature map through a
forward hook), so tests/test_oracle_golden.py can check that oracle/nets.py -- fed the BN-FOLDED weights -- reproduces them:
flatten order, LayerNorm, head slicing and view order are then the reference's, not a restatement's.
"""
import os
import sys
import tempfile

import numpy as np

REF = os.environ.get("ADAS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ufldnet_params as UP  # noqa: E402

TV_STUB = '''
import torch
from torch import nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], 2)

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        seq = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        seq += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)


def resnet18(pretrained=False, **kw):
    assert not pretrained
    return ResNet([2, 2, 2, 2])


def resnet34(pretrained=False, **kw):
    assert not pretrained
    return ResNet([3, 4, 6, 3])
'''


def install_stub():
    d = tempfile.mkdtemp(prefix="adas_tvstub_")
    os.makedirs(os.path.join(d, "torchvision"))
    open(os.path.join(d, "torchvision", "__init__.py"), "w").write("from . import models\n")
    open(os.path.join(d, "torchvision", "models.py"), "w").write(TV_STUB)
    sys.path.insert(0, d)
    # the package __init__ of TrafficLaneDetector imports cv2/onnxruntime-backed detectors: import the export library alone
    sys.path.insert(1, os.path.join(REF, "TrafficLaneDetector", "ufldDetector"))


def load_state(net, sd):
    import torch
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    own = net.state_dict()
    for k in own:
        if k.endswith("num_batches_tracked"):
            tsd[k] = own[k]
    missing, unexpected = net.load_state_dict(tsd, strict=True), None
    net.eval()
    return net


def run_case(tag, kind, depth, kw):
    import torch
    if kind == "v2":
        from exportLib.ultrafastLaneV2.model_culane import parsingNet
        nl = kw.get("lanes", 4)
        net = parsingNet(pretrained=False, backbone=depth, num_grid_row=kw["grid_row"], num_cls_row=kw["cls_row"],
                         num_grid_col=kw["grid_col"], num_cls_col=kw["cls_col"], num_lane_on_row=nl, num_lane_on_col=nl, use_aux=False,
                         input_height=kw["in_h"], input_width=kw["in_w"], fc_norm=kw["fc_norm"])
        sd = UP.ufldv2_state(UP.SEED, depth, **kw)
        h, w = kw["in_h"], kw["in_w"]
    else:
        from exportLib.ultrafastLane.model import parsingNet
        net = parsingNet(size=(288, 800), pretrained=False, backbone=depth, cls_dim=(kw["griding_num"] + 1, kw["cls_per_lane"], 4),
                         use_aux=False)
        sd = UP.ufld1_state(UP.SEED, depth, **kw)
        h, w = 288, 800
    load_state(net, sd)
    taps = {}
    net.pool.register_forward_hook(lambda m, i, o: taps.__setitem__("pool", o.detach()))
    x = UP.lane_frame(UP.SEED + 1, h, w)
    with torch.no_grad():
        y = net(torch.from_numpy(x))
    outs = [y[k] for k in ("loc_row", "loc_col", "exist_row", "exist_col")] if kind == "v2" else [y]
    outs = [o.numpy() for o in outs] + [taps["pool"].numpy()]
    rec = {}
    for i, o in enumerate(outs):
        name = "pool" if i == len(outs) - 1 else "out%d" % i
        flat = o.reshape(-1).astype(np.float32)
        rec[f"{tag}_{name}_shape"] = np.asarray(o.shape, np.int64)
        rec[f"{tag}_{name}_sample"] = flat[UP.sample_idx(flat.size)]
        rec[f"{tag}_{name}_sum"] = np.float64(flat.astype(np.float64).sum())
        rec[f"{tag}_{name}_abssum"] = np.float64(np.abs(flat.astype(np.float64)).sum())
    rec[f"{tag}_n_outputs"] = np.int64(len(outs) - 1)
    print(tag, [tuple(o.shape) for o in outs], "abs-mean out0 %.4f" % float(np.abs(outs[0]).mean()), flush=True)
    return rec


def main():
    install_stub()
    import torch
    torch.manual_seed(0)
    rec = {}
    for tag, kind, depth, kw in UP.CASES:
        rec.update(run_case(tag, kind, depth, kw))
    rec["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(HERE, "ufld_net.npz"), **rec)
    print("wrote ufld_net.npz,", os.path.getsize(os.path.join(HERE, "ufld_net.npz")), "bytes")


if __name__ == "__main__":
    main()
