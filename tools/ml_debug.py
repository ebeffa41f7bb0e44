"""GPU scratch tool (round 5): the multi-layer launch against the per-layer launches, layer by layer.
  python tools/ml_debug.py diff  [model] [batch] [prec]   -- first differing member layers (count, max |diff|)
  python tools/ml_debug.py time  [model] [batch] [prec]   -- segment times vs the summed per-layer times of their members
Environment knobs of the launch (ADAS_ML_ORDER / ADAS_ML_GRID / ADAS_ML_ONLY / ADAS_ML_MIN_LAYERS) apply."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
load_pkg()
import netutil
CE = importlib.import_module("adas_amd.coreEngine")
L = importlib.import_module("adas_amd._lib")


def engine(path, prec, batch, mode):      # "ml": multi-layer persistent launches; "group": grouped independent layers (the default path); "plain"
    os.environ["ADAS_ML"] = "1" if mode == "ml" else "0"
    if mode == "plain":
        os.environ["ADAS_NO_GROUP"] = "1"        # the reference engine: every layer its own launch
    else:
        os.environ.pop("ADAS_NO_GROUP", None)
    e = CE.HipEngine(path, precision=prec, max_batch=batch)
    e.prepare(batch)
    return e


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "diff"
    name = sys.argv[2] if len(sys.argv) > 2 else "yolov8n"
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    prec = sys.argv[4] if len(sys.argv) > 4 else "fp16"
    path, W, g = netutil.model(name)
    a, b = engine(path, prec, batch, os.environ.get("ML_DEBUG_A", "ml")), engine(path, prec, batch, "plain")
    n = a.stats()["num_layers"]
    x0 = netutil.coco_like_frames(2)
    x = np.ascontiguousarray(np.stack([np.roll(x0[i % 2], (13 * i, 29 * i), (1, 2)) for i in range(batch)])).astype(np.float32)
    print(name, prec, batch, a.ml_info(batch), "launches", a.launch_count(batch), "vs", b.launch_count(batch))
    if mode == "diff":
        ya, yb = a.engine_inference(x), b.engine_inference(x)
        print("status", a.ml_status(batch), "head equal", all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(ya, yb)))
        shown = 0
        for i in range(n):
            k = a.layer_kernel(i, batch)
            if not k.startswith(("conv_ml_kernel", "(in the multi-layer", "conv_halo_group_kernel", "(in the grouped")):
                continue
            u, v = a.fetch_activation(i, batch), b.fetch_activation(i, batch)
            nd = int((u != v).sum())
            print("%3d %-28s %-34s %-22s diff %9d / %9d  max %.3e" % (i, a.layer_info(i)[0], b.layer_kernel(i, batch), k[:22], nd, u.size, float(np.abs(u - v).max())))
            if nd:
                w = np.argwhere(u != v)
                print("      first:", w[:3].tolist(), "frames with diffs:", sorted(set(w[:, 0].tolist()))[:10], "channels:", sorted(set(w[:, 1].tolist()))[:12])
                shown += 1
                if shown >= 4:
                    break
    elif mode == "prof":   # needs ADAS_LIB=<library built with -DADAS_ML_PROF>
        import ctypes as C
        dx = L.DeviceBuffer.from_array(x)
        for _ in range(3):
            a.infer_device(dx.ptr, batch)
        names = ["ticket", "item+head", "dep wait", "desc+tile", "drain+barrier", "publish"]
        ta = a.profile(dx.ptr, batch, iters=5)
        segs = [i for i in range(n) if a.layer_kernel(i, batch).startswith("conv_ml_kernel")]
        a.infer_device(dx.ptr, batch)
        for k, si in enumerate(segs):
            w = (C.c_uint32 * 16)()
            L.check(L.lib().adas_engine_ml_counters(a.handle, batch, k, w))
            v = np.frombuffer(w, np.uint64, 7, 8)
            items = int(v[6])
            tot = float(v[:6].sum())
            print("launch %d at %-24s %.4f ms, %d items, %.0f cycles per item (thread 0): " % (k, a.layer_info(si)[0], ta[si][3], items, tot / max(items, 1)) +
                  ", ".join("%s %.0f (%.0f %%)" % (nm, float(c) / max(items, 1), 100.0 * float(c) / max(tot, 1.0)) for nm, c in zip(names, v[:6])))
    else:
        dx = L.DeviceBuffer.from_array(x)
        ta, tb = a.profile(dx.ptr, batch, iters=10), b.profile(dx.ptr, batch, iters=10)
        ta, tb = a.profile(dx.ptr, batch, iters=20), b.profile(dx.ptr, batch, iters=20)
        seg, acc = None, {}
        for i in range(n):
            k = a.layer_kernel(i, batch)
            if k.startswith(("conv_ml_kernel", "conv_halo_group_kernel")):
                seg = i
                acc[seg] = [k, ta[i][3], 0.0, 0, 0.0]
            if k.startswith(("conv_ml_kernel", "(in the multi-layer", "conv_halo_group_kernel", "(in the grouped")):
                acc[seg][2] += tb[i][3]; acc[seg][3] += 1; acc[seg][4] += ta[i][1] * batch
        for s, (k, ms, ref, cnt, fl) in acc.items():
            print("segment at %3d %-28s %-26s  %.4f ms  vs per-layer sum %.4f ms  (%d layers, %.1f GFLOP, %.0f TF/s)" % (s, a.layer_info(s)[0], k, ms, ref, cnt, fl / 1e9, fl / ms / 1e9))
        print("whole net: ml %.4f ms, per-layer %.4f ms" % (sum(t[3] for t in ta), sum(t[3] for t in tb)))
    a.close(); b.close()


main()
