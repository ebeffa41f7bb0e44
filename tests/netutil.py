"""Helpers for the network parity tests: build a model container + its named weights, frames."""
import importlib, os, tempfile
import numpy as np
from conftest import load_pkg

load_pkg()
M = importlib.import_module("adas_amd.models")
_cache = {}


def model(name, seed=0, **kw):
    """-> (path to .hipm, weights dict name->ndarray, Graph)."""
    key = (name, seed, tuple(sorted(kw.items())))
    if key not in _cache:
        ws = M.SynthWeights(seed, gain=M.synth_gain(name))
        g = M.build(name, wsrc=ws, **kw)
        d = os.environ.get("ADAS_MODEL_DIR") or tempfile.gettempdir()
        path = os.path.join(d, f"adas_{name}_{seed}_{abs(hash(key)) % 10**8}.hipm")
        g.save(path)
        _cache[key] = (path, dict(ws.store), g)
    return _cache[key]


def coco_like_frames(n, h=640, w=640, seed=1):
    """grey-noise background + random filled rectangles, as NCHW fp32 in [0,1]."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 3, h, w), np.float32)
    for i in range(n):
        img = rng.normal(114, 20, (h, w, 3)).clip(0, 255)
        for _ in range(rng.integers(5, 40)):
            x0, y0 = rng.integers(0, w - 20), rng.integers(0, h - 20)
            x1, y1 = min(w, x0 + rng.integers(10, 200)), min(h, y0 + rng.integers(10, 200))
            img[y0:y1, x0:x1] = rng.integers(0, 255, 3)
        out[i] = (img.astype(np.uint8).astype(np.float32) / 255.0).transpose(2, 0, 1)
    return out


def lane_frames(n, h=320, w=1600, seed=2):
    rng = np.random.default_rng(seed)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    x = rng.integers(0, 255, (n, 3, h, w)).astype(np.float32) / 255.0
    # low-frequency structure so activations are not pure noise
    x = 0.5 * x + 0.5 * np.repeat(np.repeat(rng.uniform(0, 1, (n, 3, h // 16, w // 16)).astype(np.float32), 16, 2), 16, 3)
    return ((x - mean) / std).astype(np.float32)
