"""CPU study: how a per-layer fp16 rounding propagates through the synthetic YOLOv8 nets, by weight gain.
Emulation: weights rounded to half, every conv output rounded to half (fp32 accumulate), head logits kept fp32."""
import sys, os, importlib
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nets
import netutil
M = netutil.M

def run(scale, gain, emul, x, sharpen=1.0):
    ws = M.SynthWeights(0, gain=gain)
    M.build("yolov8" + scale, wsrc=ws)
    W = dict(ws.store)
    for i in range(3):
        W[f"model.22.cv3.{i}.2.weight"] = W[f"model.22.cv3.{i}.2.weight"] * np.float32(sharpen)
    taps = {}
    nets.EMULATE = emul
    out = nets.yolov8_forward(x, W, scale, taps=taps)
    nets.EMULATE = None
    return out, taps

def ws_bias(scale, gain, i):
    ws = M.SynthWeights(0, gain=gain)
    M.build("yolov8" + scale, wsrc=ws)
    return ws.store[f"model.22.cv3.{i}.2.bias"]

if __name__ == "__main__":
    scale = sys.argv[1] if len(sys.argv) > 1 else "n"
    x = netutil.coco_like_frames(2, seed=11)
    for gain in [float(g) for g in sys.argv[2:]] or [1.15, 1.0, 0.9, 0.8]:
        a, ta = run(scale, gain, None, x)
        b, tb = run(scale, gain, "fp16", x)
        r = lambda u, v: float(np.linalg.norm(u - v) / (np.linalg.norm(v) + 1e-30))
        bias = np.concatenate([np.repeat(ws_bias(scale, gain, i)[:, None], n, 1) for i, n in enumerate((6400, 1600, 400))], 1)
        sigz = (ta["cls_logits"].numpy() - bias[None]).std()
        dz = (tb["cls_logits"] - ta["cls_logits"]).numpy()
        print("  logit signal std %.4f  err rms %.2e max %.2e  -> err/signal %.2e ; box err rms %.2e px" % (sigz, np.sqrt((dz ** 2).mean()), np.abs(dz).max(), np.sqrt((dz ** 2).mean()) / sigz, np.sqrt(((a[:, :4] - b[:, :4]) ** 2).mean())))
        print("gain %.2f  p3 rel %.2e (|ref|max %.2f rms %.3f)  p5 rel %.2e (rms %.3f) cls_logit maxabs %.2e (std %.2f)  head cls maxabs %.2e box maxabs %.3e px" % (
            gain, r(tb["p3"].numpy(), ta["p3"].numpy()), ta["p3"].abs().max(), ta["p3"].pow(2).mean().sqrt(), r(tb["p5"].numpy(), ta["p5"].numpy()), ta["p5"].pow(2).mean().sqrt(),
            (tb["cls_logits"] - ta["cls_logits"]).abs().max(), ta["cls_logits"].std(), np.abs(a[:, 4:] - b[:, 4:]).max(), np.abs(a[:, :4] - b[:, :4]).max()))
