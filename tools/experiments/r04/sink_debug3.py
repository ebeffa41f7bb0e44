import sys, os, importlib, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import load_pkg
load_pkg()
L = importlib.import_module("adas_amd._lib"); CE = importlib.import_module("adas_amd.coreEngine"); PP = importlib.import_module("adas_amd.postproc"); M = importlib.import_module("adas_amd.models")
PL = importlib.import_module("adas_amd.pipeline")
import netutil, bench, tempfile
S = 2
cam = bench.cam_frames(S, 90)
seam = np.concatenate([importlib.import_module("oracle.preprocess").yolo_prepare_input(f, (640, 640)) for f in cam])
det_path, _, _ = bench.build_detector(M, CE, "yolov7-tiny", seam, tempfile.gettempdir(), "sinkdbg", target_per_frame=60.0)
lane_path, _, _ = netutil.model("ufldv2_res18")
A = 25200
def views(post):
    pc, pk = C.c_void_p(), C.c_void_p()
    L.check(L.lib().adas_yolo_post_scan_views(C.c_void_p(post.h), C.byref(pc), C.byref(pk)))
    return pc.value, pk.value
def grab(ptr, shape, dt):
    a = np.empty(shape, dt)
    L.check(L.lib().adas_memcpy_d2h(L.ptr(a), C.c_void_p(ptr), a.nbytes))
    return a
for env in ("0", "1"):
    os.environ["ADAS_NO_DETECT_SINK"] = env
    p = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="fp16", src_hw=(720, 1280), use_graph=False)
    dc = L.DeviceBuffer.from_array(cam)
    p.step_frames(dc.ptr, (720, 1280), 0.6); p.sync(); L.check(L.lib().adas_synchronize())
    pc, pk = views(p.post)
    print("NO_SINK", env, "post.h %x best_conf %x best_cls %x" % (p.post.h, pc, pk))
    conf = grab(pc, (S, A), np.float32); cls = grab(pk, (S, A), np.int32)
    head = grab(p.det.output_device_ptr(0), (S, A, 85), np.float32)
    r = [PP.YoloPost.fetch(p.post, s) for s in range(S)]
    print("   cands", [len(x["cand_anchor"]) for x in r], "best_conf max %.4f n>0.4 %d; head obj max %.4f cls max %.4f, head-derived n>0.4 %d" % (
        conf.max(), (conf > 0.4).sum(), head[..., 4].max(), head[..., 5:].max(), ((head[..., 5:] * head[..., 4:5]).max(-1) > 0.4).sum()))
    dc.free(); p.close()
