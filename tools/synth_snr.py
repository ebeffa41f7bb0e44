"""CPU study: anchor-to-anchor signal of the best class logit vs its fp16-emulation error, by synthetic gain."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nets
import netutil, bench
from oracle import preprocess
M = netutil.M

def run(fam, scale, gain, x):
    name = fam + scale
    ws = M.SynthWeights(0, gain=gain)
    M.build(name, wsrc=ws)
    W = dict(ws.store)
    head = "model.23.one2one_cv3" if fam == "yolov10" else "model.22.cv3"
    fwd = (lambda x_, W_, sc, taps=None: nets.detector_forward(name, x_, W_, taps=taps))
    bias = np.concatenate([np.repeat(W[f"{head}.{i}.2.bias"][:, None], n, 1) for i, n in enumerate((6400, 1600, 400))], 1)
    res = {}
    for emu in (None, "fp16"):
        nets.EMULATE = emu
        taps = {}
        out = fwd(x, W, scale, taps=taps)
        nets.EMULATE = None
        res[emu] = ((taps["cls_logits"].numpy() - bias[None]).max(1), out, taps["p3"].numpy())
    b0, b1 = res[None][0], res["fp16"][0]
    sig = b0.std(axis=1).mean()
    err = np.sqrt(((b1 - b0) ** 2).mean())
    print("%s gain %.2f: best-logit sigma over anchors %.4e  fp16 err rms %.3e max %.3e  -> SNR %.0f ; p3 rms %.3f rel %.2e ; box err max %.3e px" % (
        name, gain, sig, err, np.abs(b1 - b0).max(), sig / err, np.sqrt((res[None][2] ** 2).mean()),
        np.linalg.norm(res["fp16"][2] - res[None][2]) / np.linalg.norm(res[None][2]), np.abs(res["fp16"][1][:, :4] - res[None][1][:, :4]).max()), flush=True)

if __name__ == "__main__":
    fam, scale = sys.argv[1], sys.argv[2]
    cams = bench.cam_frames(2, 10)
    x = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for f in cams])
    for g in sys.argv[3:]:
        run(fam, scale, float(g), x)
