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
for graph in (False, True):
    for env in ("0", "1"):
        os.environ["ADAS_NO_DETECT_SINK"] = env
        p = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="fp16", src_hw=(720, 1280), use_graph=graph)
        dc = L.DeviceBuffer.from_array(cam)
        for k in range(2):
            p.step_frames(dc.ptr, (720, 1280), 0.6); p.sync()
            r = [PP.YoloPost.fetch(p.post, s) for s in range(S)]
            print("graph", graph, "NO_SINK", env, "sink flag", L.lib().adas_pipeline_detect_sink(p.h), "step", k, "cands", [len(x["cand_anchor"]) for x in r],
                  "found", [int(x.get("n_found", -1)) for x in r], "keep", [len(x["keep"]) for x in r])
        dc.free(); p.close()
