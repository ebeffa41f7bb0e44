"""GPU: YOLOv6 v3.0 n (README.md:54 lists YOLOv6; yoloDetector.py:110-124 decodes its (1, A, 5+nc) head like v5's): the transposed-conv
up-sampling (1x1 conv + depth-to-space) against torch.conv_transpose2d, the network vs the torch oracle (fp32 <= 1e-3 on tapped
activations and the head; fp16 / bf16 bounds), the drop-in YoloDetector(model_type=YOLOV6) and the fused pipeline step against the oracle
chain."""
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


@pytest.mark.parametrize("cin,cout,hw", [(64, 64, (20, 20)), (32, 32, (40, 40)), (128, 64, (7, 13)), (64, 8, (5, 3))], ids=str)
@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("fp16", 2e-3), ("bf16", 1e-2)])
def test_transposed_conv_2x2_stride2(cin, cout, hw, prec, tol):
    H, W = hw
    batch = 3
    ws = M.SynthWeights(7, gain=1.0)
    g = M.Graph("deconvunit", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, cin, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    y = g.deconv2x2(a, cout, "up")
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"deconvunit_{cin}_{cout}_{H}_{W}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(0).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("up.d2s", batch)
    a_dev = e.fetch_activation("expand", batch)
    kernels = [e.layer_kernel(i, batch) for i in range(e.stats()["num_layers"])]
    e.close(); os.remove(path)
    want = F.conv_transpose2d(torch.from_numpy(a_dev), torch.from_numpy(ws.store["up.weight"]), torch.from_numpy(ws.store["up.bias"]), stride=2).numpy()
    assert got.shape == want.shape == (batch, cout, 2 * H, 2 * W)
    assert rel_l2(got, want) <= tol, (rel_l2(got, want), kernels)
    assert "depth2space_kernel" in kernels


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_yolov6n_640_vs_oracle(tmp_path, prec):
    import bench
    x = netutil.coco_like_frames(2, seed=11)
    path, W, g = bench.build_detector(M, CE, "yolov6n", x, str(tmp_path), "v6_" + prec, target_per_frame=100.0)
    assert abs(g.n_params / 1e6 - 4.65) < 0.01
    taps = {}
    want = nets.yolov6_forward(x, W, "n", taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    assert e.get_engine_output_shape()[0] == [[1, 8400, 85]]
    got = e.engine_inference(x)[0]
    rtol = {"fp16": 5e-3, "bf16": 4e-2}
    for lname, key in (("backbone.ERBlock_5.2.cv7.block.conv", "sppf"), ("neck.Rep_p3.block.2.rbr_reparam", "p3"), ("neck.Rep_n3.block.2.rbr_reparam", "p4"),
                       ("neck.Rep_n4.block.2.rbr_reparam", "p5")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err, rel = float(np.abs(a - ref).max()), rel_l2(a, ref)
        print("yolov6n %s %-5s max|diff| %.3e  rel_l2 %.3e  max|ref| %.2f" % (prec, key, err, rel, np.abs(ref).max()))
        if prec == "fp32":
            assert err <= 1e-3 * max(1.0, float(np.abs(ref).max())), lname
        else:
            assert rel <= rtol[prec], lname
    ecls = float(np.abs(got[..., 4:] - want[..., 4:]).max())
    atol, rtol_b = {"fp32": (1e-3, 1e-5), "fp16": (0.1, 1e-2), "bf16": (1.0, 8e-2)}[prec]
    ebox = float((np.abs(got[..., :4] - want[..., :4]) / (atol + rtol_b * np.abs(want[..., :4]))).max())
    n_over = int((want[..., 5:].max(axis=-1) > 0.4).sum())
    print("yolov6n %s head: max|prob diff| %.3e  box %.3f of its bound  (%d anchors over 0.4)" % (prec, ecls, ebox, n_over))
    assert n_over >= 50 and np.all(got[..., 4] == 1.0)
    assert ecls <= {"fp32": 1e-3, "fp16": 2e-2, "bf16": 1.5e-1}[prec] and ebox <= 1.0
    kernels = {e.layer_kernel(i, 2) for i in range(e.stats()["num_layers"])}
    print(sorted(kernels))
    assert prec == "fp32" or not any("conv_igemm" in k for k in kernels), kernels
    e.close()


def test_yolov6_detector_dropin_and_pipeline_chain(tmp_path):
    import bench
    cams = bench.cam_frames(4, 80)
    seam = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for f in cams])
    path, W, g = bench.build_detector(M, CE, "yolov6n", seam, str(tmp_path), "v6d", target_per_frame=80.0, capacity=1024)
    lab = tmp_path / "coco_label.txt"
    lab.write_text("\n".join(f"class{i}" for i in range(80)))
    det = D.YoloDetector(model_path=path, model_type=D.ObjectModelType.YOLOV6, classes_path=str(lab), box_score=0.4, box_nms_iou=0.45, precision="fp32")
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
    chain = CP.OracleChain("yolov6n", W, "ufldv2_res18", Wl)
    st = CP.run_device_chain(pipe, lambda s: PP.YoloPost.fetch(pipe.post, s), lambda s: gpu_api.track_snapshot(*pipe.tracker.fetch(s)),
                             d_pool, pool, chain, 4, 2, [0, 1])
    pipe.close()
    for b in d_pool:
        b.free()
    o = st.summary()
    print("yolov6n pipeline fp32:", o)
    n = o["frames"]
    assert o["identical_candidate_sets"] == n and o["identical_survivors"] == n and o["identical_track_ids"] == o["track_states_compared"]
    assert o["lanes_within_1px"] == n and o["survivors_compared"] >= n
