"""GPU: YOLOv7-tiny (README.md:55 lists YOLOv7; yoloDetector.py:110-124 decodes its head as the v5 layout): LeakyReLU(0.1) in every conv
kernel the graph is planned onto, the network vs the torch oracle (fp32 <= 1e-3 on tapped activations and the head; fp16 / bf16
bounds), the drop-in YoloDetector(model_type=YOLOV7) and the fused pipeline step against the oracle chain with a calibrated v5-layout
head."""
import importlib

import numpy as np
import pytest

import netutil
import gpu_api
import parity_checks as pc
import chain_parity as CP
from conftest import load_pkg
from oracle import nets, preprocess, yolo_post
from test_gpu_conv import run_case

pytestmark = pytest.mark.gpu
load_pkg()
L = importlib.import_module("adas_amd._lib")
CE = importlib.import_module("adas_amd.coreEngine")
PP = importlib.import_module("adas_amd.postproc")
PL = importlib.import_module("adas_amd.pipeline")
M = importlib.import_module("adas_amd.models")
D = importlib.import_module("adas_amd.detectors")


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30))


# (cin, cout, k, s, (H, W)): the layer shapes of yolov7-tiny at 640x640 (one per kernel the planner picks) and two ragged ones
LEAKY_CASES = [(32, 32, 3, 1, (160, 160)), (64, 64, 3, 1, (80, 80)), (128, 128, 3, 1, (40, 40)), (256, 256, 3, 1, (20, 20)), (64, 128, 3, 1, (80, 80)),
               (32, 64, 3, 2, (320, 320)), (64, 128, 3, 2, (80, 80)), (128, 256, 3, 2, (40, 40)),
               (64, 32, 1, 1, (160, 160)), (128, 64, 1, 1, (80, 80)), (256, 128, 1, 1, (40, 40)), (1024, 256, 1, 1, (20, 20)), (512, 256, 1, 1, (20, 20)),
               (64, 64, 3, 1, (23, 37)), (128, 64, 1, 1, (7, 300))]


@pytest.mark.parametrize("case", LEAKY_CASES, ids=str)
def test_leaky_relu_conv_kernels(case):
    cin, cout, k, s, (H, W) = case
    for prec, tol in (("fp16", 2e-3), ("bf16", 1e-2), ("fp32", 1e-5)):
        info = {}
        rel, mx = run_case(CE, H, W, cin, cout, k, s, M.ACT_LEAKY, M.RES_NONE, prec, info=info)
        print("leaky %s %s -> %s  rel %.2e max %.2e" % (case, prec, info["kernel"], rel, mx))
        assert rel < tol, (case, prec, info, rel, mx)
        if prec != "fp32":
            assert "igemm" not in info["kernel"], info


def test_leaky_relu_in_the_batch_64_kernels():
    """At the stream counts the pipeline runs (batch 64) the planner moves the 3x3 stride-1 layers onto the persistent kernels (conv_h8,
    conv_halo_rw): their LeakyReLU instantiations, on yolov7-tiny's own layer shapes (+ the stride-2 layers at that batch)."""
    seen = set()
    for cin, cout, k, s, (H, W), batch in [(32, 32, 3, 1, (160, 160), 16), (64, 64, 3, 1, (80, 80), 64), (128, 128, 3, 1, (40, 40), 64),
                                           (256, 512, 3, 1, (20, 20), 64), (64, 128, 3, 2, (80, 80), 64), (128, 256, 3, 2, (80, 80), 64)]:
        for prec, tol in (("fp16", 2e-3), ("bf16", 1e-2)):
            info = {}
            rel, mx = run_case(CE, H, W, cin, cout, k, s, M.ACT_LEAKY, M.RES_NONE, prec, batch=batch, info=info)
            print("leaky b%d %s -> %s  rel %.2e max %.2e" % (batch, (cin, cout, k, s, H, W), info["kernel"], rel, mx))
            assert rel < tol, (cin, cout, k, s, prec, info, rel, mx)
            seen.add(info["kernel"].split("<")[0])
    assert {"conv_h8_kernel", "conv_halo_rw_kernel"} <= seen, seen


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_yolov7_tiny_640_vs_oracle(tmp_path, prec):
    import bench
    x = netutil.coco_like_frames(2, seed=11)
    path, W, g = bench.build_detector(M, CE, "yolov7-tiny", x, str(tmp_path), "v7_" + prec, target_per_frame=100.0)
    assert g.n_params == 6219709
    taps = {}
    want = nets.yolov7_tiny_forward(x, W, taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    assert e.get_engine_output_shape()[0] == [[1, 25200, 85]]
    got = e.engine_inference(x)[0]
    rtol = {"fp16": 5e-3, "bf16": 4e-2}
    for lname, key in (("model.37.conv", "sppcspc"), ("model.74.conv", "p3"), ("model.75.conv", "p4"), ("model.76.conv", "p5")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err, rel = float(np.abs(a - ref).max()), rel_l2(a, ref)
        print("yolov7-tiny %s %-7s max|diff| %.3e  rel_l2 %.3e  max|ref| %.2f" % (prec, key, err, rel, np.abs(ref).max()))
        if prec == "fp32":
            assert err <= 1e-3 * max(1.0, float(np.abs(ref).max())), lname
        else:
            assert rel <= rtol[prec], lname
    ecls = float(np.abs(got[..., 4:] - want[..., 4:]).max())
    # boxes: |diff| <= atol + rtol * |box| (wh = (2 sigmoid)^2 * anchor reaches 1e3 px: a pure pixel bound would be about the anchors)
    atol, rtol_b = {"fp32": (1e-3, 1e-5), "fp16": (0.1, 1e-2), "bf16": (1.0, 8e-2)}[prec]
    ebox = float((np.abs(got[..., :4] - want[..., :4]) / (atol + rtol_b * np.abs(want[..., :4]))).max())
    conf = want[..., 4] * want[..., 5:].max(axis=-1)
    n_over = int((conf > 0.4).sum())
    print("yolov7-tiny %s head: max|prob diff| %.3e  max box diff / (%.0e px + %.0e |box|) = %.3f  (%d anchors over 0.4)" % (prec, ecls, atol, rtol_b, ebox, n_over))
    assert n_over >= 50
    assert ecls <= {"fp32": 1e-3, "fp16": 2e-2, "bf16": 1.5e-1}[prec] and ebox <= 1.0
    kernels = {e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])}
    print(sorted(kernels))
    assert prec == "fp32" or (not any("conv_igemm" in k for k in kernels) and "detect_v5_fused_kernel" in kernels), kernels
    e.close()


@pytest.mark.parametrize("name,hw,nc", [("yolov7-tiny", (96, 160), 80), ("yolov7-tiny", (224, 352), 3), ("yolov5n", (160, 96), 80), ("yolov5s", (64, 64), 91)], ids=str)
@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_v5_layout_detect_fused_with_its_convs(name, hw, nc, prec):
    """detect_v5_fused_kernel (the per-level 1x1 convs + sigmoid + grid / anchor decode in one launch) on ragged level sizes (cell counts
    that are no multiple of the 128-cell workgroup or of 4), 3 frames, other class counts.  The oracle's Detect is applied to the
    activations the DEVICE fed the kernel (fetched), with the weights rounded to the storage type like the packed fragments: what
    remains is the fp32 summation order, so every element of the (A, 5 + nc) head has to agree to 1e-4 (probabilities) / 1e-4 relative
    (boxes), in the oracle's (level, anchor, y, x) row order."""
    import torch
    path, W, g = netutil.model(name, imgsz=hw, nc=nc)
    x = netutil.coco_like_frames(3, hw[0], hw[1], seed=21)
    e = CE.HipEngine(path, precision=prec, max_batch=3)
    got = e.engine_inference(x)[0]
    kernels = [e.layer_kernel(i, 3) for i in range(e.stats()["num_layers"])]
    det = [o for o in g.ops if o["type"] == M.OP_DETECT_V5][0]
    head_convs = [[o for o in g.ops if o["type"] == M.OP_CONV and o["out"].buf == v.buf][0] for v in det["ins"]]
    feeders = [[o for o in g.ops if o["type"] == M.OP_CONV and o["out"].buf == hc["ins"][0].buf and o["out"].coff == hc["ins"][0].coff][0] for hc in head_convs]
    feats = [torch.from_numpy(e.fetch_activation(f["name"], 3)) for f in feeders]
    e.close()
    assert "detect_v5_fused_kernel" in kernels and kernels.count("(fused into the Detect launch)") == 3, kernels
    fmt = head_convs[0]["name"].rsplit(".", 1)[0] + ".{}"
    anchors = M.V7_TINY_ANCHORS if name.startswith("yolov7") else M.V5_ANCHORS
    nets.EMULATE = prec
    try:
        want = nets._v5_decode(feats, W, fmt, nc, anchors, hw[0])
    finally:
        nets.EMULATE = None
    assert got.shape == want.shape
    ecls = float(np.abs(got[..., 4:] - want[..., 4:]).max())
    ebox = float((np.abs(got[..., :4] - want[..., :4]) / (1e-3 + 1e-4 * np.abs(want[..., :4]))).max())
    print("%s %s %s nc=%d: max|prob diff| %.3e, box %.3f of its bound" % (name, hw, prec, nc, ecls, ebox))
    assert ecls <= 1e-4 and ebox <= 1.0


def test_yolov7_detector_dropin_and_pipeline_chain(tmp_path):
    import bench
    cams = bench.cam_frames(4, 79)
    seam = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for f in cams])
    path, W, g = bench.build_detector(M, CE, "yolov7-tiny", seam, str(tmp_path), "v7d", target_per_frame=80.0, capacity=1024)
    lab = tmp_path / "coco_label.txt"
    lab.write_text("\n".join(f"class{i}" for i in range(80)))
    det = D.YoloDetector(model_path=path, model_type=D.ObjectModelType.YOLOV7, classes_path=str(lab), box_score=0.4, box_nms_iou=0.45, precision="fp32")
    eng = CE.OnnxEngine(path, precision="fp32")
    lb = yolo_post.letterbox_params((720, 1280), (640, 640))
    n_box = 0
    for f in cams[:2]:
        det.DetectFrame(f)
        head = eng.engine_inference(preprocess.yolo_prepare_input(f, (640, 640)))[0][0]
        want = yolo_post.detect_post(head, lb, "yolov5", 0.4, 0.45)
        pc.check_yolo(det._last, want)
        n_box += len(want["conf"])
    assert n_box > 0
    det.close(); eng.close()
    lane_path, Wl, gl = netutil.model("ufldv2_res18")
    pool = [cams[:2], cams[2:]]
    pipe = PL.AdasPipeline(path, lane_path, n_streams=2, precision="fp32", src_hw=(720, 1280), head_layout=L.HEAD_V5, use_graph=True, max_candidates=1024)
    d_pool = [L.DeviceBuffer.from_array(np.ascontiguousarray(p)) for p in pool]
    chain = CP.OracleChain("yolov7-tiny", W, "ufldv2_res18", Wl)
    st = CP.run_device_chain(pipe, lambda s: PP.YoloPost.fetch(pipe.post, s), lambda s: gpu_api.track_snapshot(*pipe.tracker.fetch(s)),
                             d_pool, pool, chain, 4, 2, [0, 1])
    pipe.close()
    for b in d_pool:
        b.free()
    o = st.summary()
    print("yolov7-tiny pipeline fp32:", o)
    n = o["frames"]
    assert o["identical_candidate_sets"] == n and o["identical_survivors"] == n and o["identical_track_ids"] == o["track_states_compared"]
    assert o["lanes_within_1px"] == n and o["survivors_compared"] >= n
