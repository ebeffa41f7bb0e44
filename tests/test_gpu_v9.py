"""GPU: YOLOv9t (GELAN-t; README.md:57 lists YOLOv9, yoloDetector.py:114,121 decodes its head like v8's): the average-pool kernel,
network vs the torch oracle (fp32 <= 1e-3 on tapped activations and the head; fp16 / bf16 bounds on a calibrated head), the drop-in
YoloDetector(model_type=YOLOV9) and the fused pipeline step against the oracle chain."""
import importlib, os, tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import netutil
import gpu_api
import parity_checks as pc
import chain_parity as CP
from conftest import load_pkg
from oracle import nets, preprocess, yolo_post

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


@pytest.mark.parametrize("k,s,p,c,hw", [(2, 1, 0, 32, (160, 160)), (2, 1, 0, 96, (23, 37)), (3, 2, 1, 64, (40, 56)), (2, 2, 0, 16, (20, 20))], ids=str)
@pytest.mark.parametrize("prec,tol", [("fp32", 1e-6), ("fp16", 1e-3), ("bf16", 8e-3)])
def test_average_pool_kernel(k, s, p, c, hw, prec, tol):
    H, W = hw
    batch = 2
    ws = M.SynthWeights(5, gain=1.0)
    g = M.Graph("apunit", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, c, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    y = g.avgpool(a, k, s, p, name="test")
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"apunit_{k}_{s}_{p}_{c}_{H}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(0).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("test", batch)
    a_dev = e.fetch_activation("expand", batch)
    e.close(); os.remove(path)
    want = F.avg_pool2d(torch.from_numpy(a_dev), k, s, p, False, True).numpy()
    assert got.shape == want.shape and rel_l2(got, want) <= tol, rel_l2(got, want)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_yolov9t_640_vs_oracle(tmp_path, prec):
    import bench
    x = netutil.coco_like_frames(2, seed=11)
    path, W, g = bench.build_detector(M, CE, "yolov9t", x, str(tmp_path), "v9_" + prec, target_per_frame=100.0)
    assert abs(g.flops / 1e9 - 8.23) < 0.05
    taps = {}
    want = nets.yolov9t_forward(x, W, taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    assert e.get_engine_output_shape()[0] == [[1, 84, 8400]]
    got = e.engine_inference(x)[0]
    rtol = {"fp16": 5e-3, "bf16": 4e-2}
    for lname, key in (("model.9.cv5.conv", "sppelan"), ("model.15.cv4.conv", "p3"), ("model.18.cv4.conv", "p4"), ("model.21.cv4.conv", "p5")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err, rel = float(np.abs(a - ref).max()), rel_l2(a, ref)
        print("yolov9t %s %-7s max|diff| %.3e  rel_l2 %.3e  max|ref| %.2f" % (prec, key, err, rel, np.abs(ref).max()))
        if prec == "fp32":
            assert err <= 1e-3 * max(1.0, float(np.abs(ref).max())), lname
        else:
            assert rel <= rtol[prec], lname
    ecls = float(np.abs(got[:, 4:] - want[:, 4:]).max())
    ebox = float(np.abs(got[:, :4] - want[:, :4]).max())
    n_over = int((want[:, 4:].max(axis=1) > 0.4).sum())
    print("yolov9t %s head: max|prob diff| %.3e  max|box diff| %.3e px  (%d anchors over 0.4)" % (prec, ecls, ebox, n_over))
    assert n_over >= 50
    if prec == "fp32":
        assert ecls <= 1e-3 and ebox <= 1e-3 * max(1.0, float(np.abs(want[:, :4]).max()))
    else:
        assert ecls <= {"fp16": 2e-2, "bf16": 1.5e-1}[prec] and ebox <= {"fp16": 0.1, "bf16": 1.0}[prec]
    kernels = {e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])}
    assert "avgpool_kernel" in kernels and not any("conv_igemm" in k for k in kernels) or prec == "fp32", kernels
    e.close()


@pytest.mark.parametrize("name", ["yolov9s", "yolov9c"])
@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_yolov9s_c_vs_oracle(prec, name):
    """YOLOv9s = the t graph with doubled widths (7.2 M parameters); YOLOv9c = GELAN-c with ADown (25.4 M): 384x640 input, 2 frames,
    tapped activations and the head."""
    path, W, g = netutil.model(name, imgsz=(384, 640))
    x = netutil.coco_like_frames(2, 384, 640, seed=12)
    taps = {}
    want = nets.detector_forward(name, x, W, taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    got = e.engine_inference(x)[0]
    for lname, key in (("model.9.cv5.conv", "sppelan"), ("model.15.cv4.conv", "p3"), ("model.21.cv4.conv", "p5")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err, rel = float(np.abs(a - ref).max()), rel_l2(a, ref)
        print(name + " %s %-7s max|diff| %.3e  rel_l2 %.3e  max|ref| %.2f" % (prec, key, err, rel, np.abs(ref).max()))
        assert (err <= 1e-3 * max(1.0, float(np.abs(ref).max()))) if prec == "fp32" else (rel <= 5e-3), lname
    ecls, ebox = float(np.abs(got[:, 4:] - want[:, 4:]).max()), float(np.abs(got[:, :4] - want[:, :4]).max())
    print(name + " %s head: max|prob diff| %.3e  max|box diff| %.3e px" % (prec, ecls, ebox))
    assert got.shape == want.shape == (2, 84, 5040)
    if prec == "fp32":
        assert ecls <= 1e-3 and ebox <= 1e-3 * max(1.0, float(np.abs(want[:, :4]).max()))
    else:
        assert ecls <= 2e-2 and ebox <= 0.25
    e.close()


def test_yolov9_detector_dropin_and_pipeline_chain(tmp_path):
    import bench
    cams = bench.cam_frames(4, 78)
    seam = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for f in cams])
    path, W, g = bench.build_detector(M, CE, "yolov9t", seam, str(tmp_path), "v9d", target_per_frame=80.0, capacity=1024)
    lab = tmp_path / "coco_label.txt"
    lab.write_text("\n".join(f"class{i}" for i in range(80)))
    det = D.YoloDetector(model_path=path, model_type=D.ObjectModelType.YOLOV9, classes_path=str(lab), box_score=0.4, box_nms_iou=0.45, precision="fp32")
    eng = CE.OnnxEngine(path, precision="fp32")
    lb = yolo_post.letterbox_params((720, 1280), (640, 640))
    for f in cams[:2]:
        det.DetectFrame(f)
        head = eng.engine_inference(preprocess.yolo_prepare_input(f, (640, 640)))[0][0]
        want = yolo_post.detect_post(head, lb, "yolov8", 0.4, 0.45)
        pc.check_yolo(det._last, want)
    det.close(); eng.close()
    lane_path, Wl, gl = netutil.model("ufldv2_res18")
    pool = [cams[:2], cams[2:]]
    pipe = PL.AdasPipeline(path, lane_path, n_streams=2, precision="fp32", src_hw=(720, 1280), use_graph=True, max_candidates=1024)
    d_pool = [L.DeviceBuffer.from_array(np.ascontiguousarray(p)) for p in pool]
    chain = CP.OracleChain("yolov9t", W, "ufldv2_res18", Wl)
    st = CP.run_device_chain(pipe, lambda s: PP.YoloPost.fetch(pipe.post, s), lambda s: gpu_api.track_snapshot(*pipe.tracker.fetch(s)),
                             d_pool, pool, chain, 4, 2, [0, 1])
    pipe.close()
    for b in d_pool:
        b.free()
    o = st.summary()
    print("yolov9t pipeline fp32:", o)
    n = o["frames"]
    assert o["identical_candidate_sets"] == n and o["identical_survivors"] == n and o["identical_track_ids"] == o["track_states_compared"]
    assert o["lanes_within_1px"] == n and o["survivors_compared"] >= n
