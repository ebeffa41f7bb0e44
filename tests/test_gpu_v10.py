"""GPU: YOLOv10n -- the reference's shipped default detector (demo.py:24-30 `yolov10n-coco_fp16.trt`, ObjectModelType.YOLOV10,
decoded as a v8-layout head: yoloDetector.py:114,121).  Network vs the torch oracle (fp32 <= 1e-3 on tapped activations incl. the
PSA attention block, and on the head; fp16 / bf16 bounded on a calibrated head), the depth-wise and attention kernels on their own,
and the drop-in YoloDetector(model_type=YOLOV10) frame -> RectInfo path and the fused pipeline step against the oracle chain."""
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


@pytest.mark.parametrize("k,s,c,act,res", [(3, 1, 64, M.ACT_SILU, False), (3, 2, 128, M.ACT_NONE, False), (7, 1, 256, M.ACT_SILU, False),
                                           (3, 1, 80, M.ACT_SILU, True), (3, 1, 16, M.ACT_NONE, True)], ids=str)
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("fp16", 3e-3), ("bf16", 2e-2)])
def test_depthwise_conv_kernel(k, s, c, act, res, prec, tol):
    H, W, batch = 23, 37, 3
    ws = M.SynthWeights(5, gain=1.0)
    g = M.Graph("dwunit", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, c, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    y = g.dwconv(a, k, s, "test", act=act, res=a if (res and s == 1) else None)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"dwunit_{k}_{s}_{c}_{act}_{int(res)}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(0).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("test", batch)
    assert "dwconv_kernel" in e.layer_kernel(e.layer_index("test"), batch)
    e.close(); os.remove(path)
    Wt = {n: torch.from_numpy(v) for n, v in ws.store.items()}
    with torch.no_grad():
        a_ = F.silu(F.conv2d(torch.from_numpy(xin), Wt["expand.weight"], Wt["expand.bias"]))
        y_ = F.conv2d(a_, Wt["test.weight"], Wt["test.bias"], stride=s, padding=k // 2, groups=c)
        y_ = F.silu(y_) if act == M.ACT_SILU else y_
        if res and s == 1:
            y_ = y_ + a_
    want = y_.numpy()
    assert got.shape == want.shape
    assert rel_l2(got, want) <= tol, (rel_l2(got, want), float(np.abs(got - want).max()))


@pytest.mark.parametrize("k,s,c", [(3, 1, 64), (3, 2, 144), (5, 1, 240), (5, 2, 96), (7, 1, 256)], ids=str)
@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp16x3"])
def test_depthwise_strip_form_is_bit_identical_to_the_plain_kernel(tmp_path, monkeypatch, k, s, c, prec):
    """dwconv_strip_kernel (4 outputs per thread, a tap row of weights in registers) accumulates every output's taps in the plain kernel's
    (row, column) order: the same bits, on widths that are not a multiple of the strip (37) and at the image borders."""
    H, W, batch = 23, 37, 3
    ws = M.SynthWeights(5, gain=1.0)
    g = M.Graph("dwstrip", 3, H, W, ws)
    x, c3 = g.input()
    a = g.conv(x, c, 1, 1, "expand", act=M.ACT_SILU, true_cin=c3)
    y = g.dwconv(a, k, s, "test", act=M.ACT_SILU, res=a if s == 1 else None)
    z = g.conv(y, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = str(tmp_path / "dwstrip.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(0).uniform(0, 1, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    strip = e.fetch_activation("test", batch).copy()
    monkeypatch.setenv("ADAS_NO_DW_STRIP", "1")
    e.engine_inference(xin)
    plain = e.fetch_activation("test", batch).copy()
    e.close()
    assert np.abs(strip).max() > 1e-3
    np.testing.assert_array_equal(strip, plain)


@pytest.mark.parametrize("hw,nh", [((20, 20), 2), ((12, 20), 2), ((9, 7), 1), ((20, 20), 4)], ids=str)
@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("fp16", 3e-3), ("bf16", 2e-2)])
def test_attention_kernel(hw, nh, prec, tol):
    """PSA attention core (ultralytics Attention.forward between its qkv and proj convolutions): N = H*W tokens in chunks of 64 keys
    (a ragged last chunk), several heads, against torch softmax attention on the same qkv tensor."""
    H, W = hw
    kd, hd, batch = 32, 64, 2
    c = nh * hd
    ws = M.SynthWeights(9, gain=3.0)                       # large logits: the softmax is far from uniform
    g = M.Graph("attnunit", 3, H, W, ws)
    x, c3 = g.input()
    qkv = g.conv(x, nh * (2 * kd + hd), 1, 1, "qkv", act=M.ACT_NONE, true_cin=c3)
    att = g.attention(qkv, nh, kd, hd, "test")
    z = g.conv(att, 8, 1, 1, "tap", act=M.ACT_NONE, f32_out=True)
    g.output(z, 0, [1, z.h * z.w * 8], "o")
    path = os.path.join(tempfile.gettempdir(), f"attnunit_{H}_{W}_{nh}.hipm")
    g.save(path)
    e = CE.HipEngine(path, prec, batch)
    xin = np.random.default_rng(1).uniform(-2, 2, (batch, 3, H, W)).astype(np.float32)
    e.engine_inference(xin)
    got = e.fetch_activation("test", batch)
    qkv_dev = e.fetch_activation("qkv", batch)              # the device's own (possibly 16-bit) qkv: isolates the attention kernel
    e.close(); os.remove(path)
    with torch.no_grad():
        t = torch.from_numpy(qkv_dev)
        B, N = batch, H * W
        q, k_, v = t.view(B, nh, 2 * kd + hd, N).split([kd, kd, hd], dim=2)
        attn = ((q.transpose(-2, -1) @ k_) * kd ** -0.5).softmax(dim=-1)
        want = (v @ attn.transpose(-2, -1)).reshape(B, c, H, W).numpy()
    assert got.shape == want.shape
    assert rel_l2(got, want) <= tol, (rel_l2(got, want), float(np.abs(got - want).max()))


@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
def test_yolov10n_640_vs_oracle(tmp_path, prec):
    import bench
    x = netutil.coco_like_frames(2, seed=11)
    path, W, g = bench.build_detector(M, CE, "yolov10n", x, str(tmp_path), "v10_" + prec, target_per_frame=100.0)
    assert abs(g.flops / 1e9 - 6.76) < 0.02
    taps = {}
    want = nets.yolov10_forward(x, W, "n", taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    assert e.get_engine_output_shape()[0] == [[1, 84, 8400]]
    got = e.engine_inference(x)[0]
    rtol = {"fp16": 5e-3, "bf16": 4e-2}
    for lname, key in (("model.10.cv2.conv", "psa"), ("model.16.cv2.conv", "p3"), ("model.19.cv2.conv", "p4"), ("model.22.cv2.conv", "p5")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err, rel = float(np.abs(a - ref).max()), rel_l2(a, ref)
        print("yolov10n %s %-4s max|diff| %.3e  rel_l2 %.3e  max|ref| %.2f" % (prec, key, err, rel, np.abs(ref).max()))
        if prec == "fp32":
            assert err <= 1e-3 * max(1.0, float(np.abs(ref).max())), lname
        else:
            assert rel <= rtol[prec], lname
    ecls = float(np.abs(got[:, 4:] - want[:, 4:]).max())
    ebox = float(np.abs(got[:, :4] - want[:, :4]).max())
    n_over = int((want[:, 4:].max(axis=1) > 0.4).sum())
    print("yolov10n %s head: max|prob diff| %.3e  max|box diff| %.3e px  (%d anchors over 0.4)" % (prec, ecls, ebox, n_over))
    assert n_over >= 50
    if prec == "fp32":
        assert ecls <= 1e-3 and ebox <= 1e-3 * max(1.0, float(np.abs(want[:, :4]).max()))
    else:
        assert ecls <= {"fp16": 1.5e-2, "bf16": 1e-1}[prec] and ebox <= {"fp16": 0.1, "bf16": 1.0}[prec]
    e.close()


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_yolov10s_vs_oracle(prec):
    """YOLOv10s (7.2 M parameters; the n yaml at width 0.5 with a C2fCIB + 7x7 depth-wise branch in backbone row 8, 4-head PSA attention on
    512 channels): 384x640 input, 2 frames, tapped activations and the head."""
    path, W, g = netutil.model("yolov10s", imgsz=(384, 640))
    x = netutil.coco_like_frames(2, 384, 640, seed=12)
    taps = {}
    want = nets.yolov10_forward(x, W, "s", taps=taps)
    e = CE.HipEngine(path, precision=prec, max_batch=2)
    got = e.engine_inference(x)[0]
    for lname, key in (("model.10.cv2.conv", "psa"), ("model.16.cv2.conv", "p3"), ("model.22.cv2.conv", "p5")):
        a = e.fetch_activation(lname, 2)
        ref = taps[key].numpy()
        err, rel = float(np.abs(a - ref).max()), rel_l2(a, ref)
        print("yolov10s %s %-4s max|diff| %.3e  rel_l2 %.3e  max|ref| %.2f" % (prec, key, err, rel, np.abs(ref).max()))
        assert (err <= 1e-3 * max(1.0, float(np.abs(ref).max()))) if prec == "fp32" else (rel <= 5e-3), lname
    ecls, ebox = float(np.abs(got[:, 4:] - want[:, 4:]).max()), float(np.abs(got[:, :4] - want[:, :4]).max())
    print("yolov10s %s head: max|prob diff| %.3e  max|box diff| %.3e px" % (prec, ecls, ebox))
    assert got.shape == want.shape == (2, 84, 5040)
    if prec == "fp32":
        assert ecls <= 1e-3 and ebox <= 1e-3 * max(1.0, float(np.abs(want[:, :4]).max()))
    else:
        assert ecls <= 2e-2 and ebox <= 0.25
    e.close()


def _frames(n, seed):
    import bench
    return bench.cam_frames(n, seed)


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
def test_yolov10_detector_dropin_and_pipeline_chain(tmp_path, prec):
    """demo.py's default configuration (ObjectModelType.YOLOV10, box_score 0.4, box_nms_iou 0.5): YoloDetector frame -> RectInfo
    against the oracle's post-processing of the engine's own head, then the fused step (YOLOv10n + UFLDv2-R18 + ByteTrack) against
    the whole oracle chain -- in both parity modes (fp32 on the f32 MFMA, fp16x3 on the 16-bit one)."""
    import bench
    cams = _frames(4, 77)
    seam = np.concatenate([preprocess.yolo_prepare_input(f, (640, 640)) for f in cams])
    path, W, g = bench.build_detector(M, CE, "yolov10n", seam, str(tmp_path), "v10d", target_per_frame=80.0, capacity=1024)
    lab = tmp_path / "coco_label.txt"
    lab.write_text("\n".join(f"class{i}" for i in range(80)))
    det = D.YoloDetector(model_path=path, model_type=D.ObjectModelType.YOLOV10, classes_path=str(lab), box_score=0.4, box_nms_iou=0.5, precision=prec)
    eng = CE.OnnxEngine(path, precision=prec)
    lb = yolo_post.letterbox_params((720, 1280), (640, 640))
    n_total = 0
    for f in cams[:2]:
        det.DetectFrame(f)
        head = eng.engine_inference(preprocess.yolo_prepare_input(f, (640, 640)))[0][0]
        want = yolo_post.detect_post(head, lb, "yolov8", 0.4, 0.5)
        pc.check_yolo(det._last, want)
        assert [r.tolist() for r in det.object_info] == [list(v) for v in want["xyxy_int"]]
        n_total += len(want["keep"])
    assert n_total >= 2
    det.close(); eng.close()
    lane_path, Wl, gl = netutil.model("ufldv2_res18")
    S = 2
    pool = [cams[:2], cams[2:]]
    pipe = PL.AdasPipeline(path, lane_path, n_streams=S, precision=prec, src_hw=(720, 1280), use_graph=True, max_candidates=1024)
    d_pool = [L.DeviceBuffer.from_array(np.ascontiguousarray(p)) for p in pool]
    chain = CP.OracleChain("yolov10n", W, "ufldv2_res18", Wl)
    st = CP.run_device_chain(pipe, lambda s: PP.YoloPost.fetch(pipe.post, s), lambda s: gpu_api.track_snapshot(*pipe.tracker.fetch(s)),
                             d_pool, pool, chain, 4, 2, list(range(S)))
    pipe.close()
    for b in d_pool:
        b.free()
    o = st.summary()
    print("yolov10n pipeline %s:" % prec, o)
    n = o["frames"]
    assert o["identical_candidate_sets"] == n and o["identical_survivors"] == n and o["identical_track_ids"] == o["track_states_compared"]
    assert o["lanes_within_1px"] == n and o["survivors_compared"] >= n
