import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import gpu_api as G, synth
from oracle import yolo_post
lbp = yolo_post.letterbox_params((720, 1280), (640, 640))
for n_hot in (20, 60, 400):
    head = synth.synth_v8_head(5, n_hot, n_hot // 3)[None]
    yp = G.PP.YoloPost(0, 8400, 80, 0.3, 0.45, lbp, 0, 1024, max_batch=1)
    buf = G.L.DeviceBuffer.from_array(head)
    for name, f in (("fetch (per array)", yp.fetch), ("fetch_dets (packed)", yp.fetch_dets)):
        for _ in range(20):
            yp.run_device(buf.ptr, 1); f(0)
        t = time.perf_counter()
        for _ in range(300):
            yp.run_device(buf.ptr, 1); r = f(0)
        dt = (time.perf_counter() - t) / 300
        print("n_hot %4d survivors %3d  %-22s run+fetch %.1f us" % (n_hot, len(r["keep"]), name, dt * 1e6))
    buf.free(); yp.close()
