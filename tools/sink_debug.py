"""Debug aid: per-anchor (best confidence, class) from the fused Detect kernel's sink against yolo_scan on the full head (same engine, same input)."""
import sys, os, importlib, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import load_pkg
load_pkg()
L = importlib.import_module("adas_amd._lib"); CE = importlib.import_module("adas_amd.coreEngine"); PP = importlib.import_module("adas_amd.postproc"); M = importlib.import_module("adas_amd.models")
import netutil
name = sys.argv[1] if len(sys.argv) > 1 else "yolov7-tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
path, W, g = netutil.model(name)
if len(sys.argv) > 3:      # a calibrated detector (class bias set for ~60 candidates per frame), like the pipeline tests use
    import bench, tempfile
    cam = bench.cam_frames(B, 90)
    seam = np.concatenate([importlib.import_module("oracle.preprocess").yolo_prepare_input(f, (640, 640)) for f in cam])
    path, W, g = bench.build_detector(M, CE, name, seam, tempfile.gettempdir(), "sinkdbg1", target_per_frame=60.0)
e = CE.HipEngine(path, "fp16", B)
x = netutil.coco_like_frames(B, seed=5) if len(sys.argv) <= 3 else seam.astype(np.float32)
dx = L.DeviceBuffer.from_array(x)
A, nc = g.meta["anchors"], g.meta["nc"]
v5 = g.meta["kind"] in ("yolov5", "yolov6", "yolov7")
lb = PP.letterbox((720, 1280), (640, 640))
post = PP.YoloPost(L.HEAD_V5 if v5 else L.HEAD_V8, A, nc, 0.001, 0.45, lb, L.NMS_REFERENCE, 1024, B)
e.infer_device(dx.ptr, B)
L.check(L.lib().adas_yolo_post_run(post.h, e.output_device_ptr(0), B, None)); L.check(L.lib().adas_synchronize())
pc, pk = C.c_void_p(), C.c_void_p()
L.check(L.lib().adas_yolo_post_scan_views(post.h, C.byref(pc), C.byref(pk)))
conf0 = np.empty((B, A), np.float32); cls0 = np.empty((B, A), np.int32)
L.check(L.lib().adas_memcpy_d2h(L.ptr(conf0), pc.value, conf0.nbytes)); L.check(L.lib().adas_memcpy_d2h(L.ptr(cls0), pk.value, cls0.nbytes))
sc = L.DeviceBuffer(B * A * 4); sk = L.DeviceBuffer(B * A * 4)
print("sink supported:", L.lib().adas_engine_detect_sink_supported(e.handle))
L.check(L.lib().adas_engine_set_detect_sink(e.handle, sc.ptr, sk.ptr))
e.infer_device(dx.ptr, B); L.check(L.lib().adas_synchronize())
L.check(L.lib().adas_engine_set_detect_sink(e.handle, None, None))
conf1 = sc.download((B, A), np.float32); cls1 = sk.download((B, A), np.int32)
print("n > 0.4: scan %d sink %d" % ((conf0 > 0.4).sum(), (conf1 > 0.4).sum()))
print(name, "A", A, "conf scan: max %.4f mean %.5f | sink: max %.4f mean %.5f" % (conf0.max(), conf0.mean(), conf1.max(), conf1.mean()))
d = np.nonzero((conf0 != conf1) | (cls0 != cls1))
print("differing anchors:", len(d[0]), "of", B * A)
for b, a in list(zip(*d))[:10]:
    print("  frame %d anchor %d: scan (%.6f, %d) sink (%.6f, %d)" % (b, a, conf0[b, a], cls0[b, a], conf1[b, a], cls1[b, a]))
