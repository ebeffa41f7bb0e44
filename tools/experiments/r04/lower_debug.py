import sys, os, importlib
import numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from conftest import load_pkg
load_pkg()
CE = importlib.import_module("adas_amd.coreEngine"); M = importlib.import_module("adas_amd.models")
OI = importlib.import_module("adas_amd.onnx_import"); OL = importlib.import_module("adas_amd.onnx_lower")
import graph_interp, onnx_emit, tempfile
from test_onnx_lower import fuse_graph
g = fuse_graph(128)
d = tempfile.mkdtemp()
onnx_emit.emit(g, d + "/fuse.onnx")
g2 = OL.lower(OI.read_onnx(d + "/fuse.onnx"), "t")
x = np.random.default_rng(3).uniform(0, 1, (2, 3, 128, 128)).astype(np.float32)
for tag, gg in (("original", g),):
    taps = {}
    want = graph_interp.run(gg, x, taps=taps)[0]
    path = gg.save(d + "/%s.hipm" % tag)
    e = CE.HipEngine(path, "fp32", 2)
    got = e.engine_inference(x)[0]
    print(tag, "head rel %.2e" % (np.linalg.norm(got - want) / np.linalg.norm(want)), got.shape)
    offs = [0, 256, 320, 336]
    for l in range(3):
        a, w = got[:, :, offs[l]:offs[l + 1]], want[:, :, offs[l]:offs[l + 1]]
        print("   level %d: box rel %.2e  cls rel %.2e   box got %s want %s" % (l, np.linalg.norm(a[:, :4] - w[:, :4]) / np.linalg.norm(w[:, :4]),
              np.linalg.norm(a[:, 4:] - w[:, 4:]) / np.linalg.norm(w[:, 4:]), a[0, :4, 0], w[0, :4, 0]))
    if tag == "lowered": break
    for i in range(e.stats()["num_layers"]):
        name = e.layer_info(i)[0]
        if name in taps:
            try:
                a = e.fetch_activation(i, 2)
            except Exception as ex:
                print("  %-22s fetch failed: %s" % (name, str(ex)[:60])); continue
            w = taps[name]
            print("  %-22s %-28s rel %.2e" % (name, e.layer_kernel(i, 2)[:28], np.linalg.norm(a - w) / (np.linalg.norm(w) + 1e-30)))
    e.close()
