"""GPU: the fused per-frame step (adas_pipeline_*: hipGraph replay, detector and lane branches forked onto two HIP
streams, everything device-resident) must give exactly what the same kernels give when driven one component at a
time through the host-facing entry points, for several streams and several frames (tracker state carries over)."""
import importlib
import numpy as np
import pytest

import netutil, gpu_api, parity_checks as pc
from conftest import load_pkg

pytestmark = pytest.mark.gpu
load_pkg()
L = importlib.import_module("adas_amd._lib")
CE = importlib.import_module("adas_amd.coreEngine")
PP = importlib.import_module("adas_amd.postproc")
PL = importlib.import_module("adas_amd.pipeline")
M = importlib.import_module("adas_amd.models")

LANE_KW = dict(in_h=160, in_w=800, num_grid_row=100, num_cls_row=36, num_grid_col=50, num_cls_col=41)
LANE_CFG = dict(grid_row=100, cls_row=36, grid_col=50, cls_col=41, row_anchor=np.linspace(0.42, 1, 36), col_anchor=np.linspace(0, 1, 41))


@pytest.mark.parametrize("use_graph,overlap", [(True, True), (True, False), (False, False)])
def test_pipeline_equals_components(tmp_path, use_graph, overlap):
    import bench
    S, steps = 3, 4
    frames = [netutil.coco_like_frames(S, seed=20 + i) for i in range(2)]
    lanes_in = [netutil.lane_frames(S, 160, 800, seed=30 + i) for i in range(2)]
    det_path, _, _ = bench.build_detector(M, CE, "yolov8n", frames[0], str(tmp_path), "p", target_per_frame=25.0)
    lane_path, _, _ = netutil.model("ufldv2_res18", **LANE_KW)
    A = importlib.import_module("adas_amd.analysis")
    Mh = A.PerspectiveTransformation((1280, 720)).M
    pipe = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="bf16", src_hw=(720, 1280), use_graph=use_graph,
                           max_candidates=512, lane_cfg=LANE_CFG, overlap=overlap, geometry=dict(bird_wh=(1280, 720), M=Mh))
    geo = PP.LaneGeometry(720, (1280, 720), Mh, True, S)
    # the step feeds the post-processing's per-anchor (best probability, class) arrays from the fused Detect kernel's registers and never
    # writes the head's class rows; the component path below writes the whole head and scans it: same candidates, bit for bit
    assert L.lib().adas_pipeline_detect_sink(pipe.h) == 1
    d_det = [L.DeviceBuffer.from_array(f) for f in frames]
    d_lane = [L.DeviceBuffer.from_array(f) for f in lanes_in]

    # component path: same engines' kernels through the host entry points, one piece at a time
    det = CE.HipEngine(det_path, "bf16", S)
    lane = CE.HipEngine(lane_path, "bf16", S)
    lb = PP.letterbox((720, 1280), (640, 640))
    post = PP.YoloPost(L.HEAD_V8, 8400, 80, 0.4, 0.45, lb, L.NMS_REFERENCE, 512, S)
    dec = PP.UfldDecode(100, 36, 50, 41, 1280, 720, LANE_CFG["row_anchor"], LANE_CFG["col_anchor"], 1, S)
    trk = PP.DeviceTracker(S, max_dets=512)

    n_det = 0
    for k in range(steps):
        i = (k // 2) % 2                                   # each frame set shown twice: tracks get confirmed, then lost
        pipe.step(d_det[i].ptr, d_lane[i].ptr)
        pipe.sync()
        heads = det.engine_inference(frames[i])[0]
        want_det = post.run_host(heads)
        want_lane = dec.run_host(lane.engine_inference(lanes_in[i]))
        geo.run(dec, True, S)
        for s in range(S):
            gg, wg = pipe.geometry.fetch(s), geo.fetch(s)     # the optional lane-geometry stage behind the decode
            assert gg["area_status"] == wg["area_status"] and gg["direction"] == wg["direction"]
            np.testing.assert_array_equal(gg["area_points"], wg["area_points"])
            for li in range(4):
                np.testing.assert_array_equal(gg["bird_points"][li], wg["bird_points"][li])
            assert gg["curvature"] == wg["curvature"] and gg["offset"] == wg["offset"]
            got = PP.YoloPost.fetch(pipe.post, s)
            for key in ("cand_anchor", "cand_conf", "cand_xywh", "keep", "xywh", "conf", "class_id", "xyxy_int"):
                np.testing.assert_array_equal(got[key], want_det[s][key], err_msg=f"step {k} stream {s} {key}")
            n_det += len(got["keep"])
            gl, gs = pipe.decode.fetch(s)
            assert (gl, gs) == want_lane[s], (k, s)
            trk.update_host(s, want_det[s]["xyxy_int"], want_det[s]["conf"], want_det[s]["class_id"])
            pc.check_track_frame(gpu_api.track_snapshot(*pipe.tracker.fetch(s)), gpu_api.track_snapshot(*trk.fetch(s)), ctx=(k, s))
    assert n_det > 10
    t = pipe.timings()
    assert t["step"] > 0
    pipe.close(); det.close(); lane.close(); post.close(); dec.close(); trk.close(); geo.close()


def test_pipeline_step_from_camera_frames(tmp_path):
    """adas_pipeline_step_frames: u8 frames -> both pre-processings inside the (graph-captured) step == pre-processing the
    frames with the stand-alone kernels and stepping from the seam tensors."""
    import bench
    S = 2
    cam = [bench.cam_frames(S, 70 + i) for i in range(2)]
    lane_path, _, _ = netutil.model("ufldv2_res18")
    det_path = M.build("yolov8n").save(str(tmp_path / "d.hipm"))
    pa = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="bf16", src_hw=(720, 1280), use_graph=True)
    pb = PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="bf16", src_hw=(720, 1280), use_graph=False)
    dt = L.DeviceBuffer(S * 3 * 640 * 640 * 4); lt = L.DeviceBuffer(S * 3 * 320 * 1600 * 4)
    import ctypes as C
    for k in (0, 1, 0):
        dc = L.DeviceBuffer.from_array(cam[k])
        pa.step_frames(dc.ptr, (720, 1280), 0.6); pa.sync()
        L.check(L.lib().adas_preprocess_yolo(dc.ptr, S, 720, 1280, dt.ptr, 640, 640, 1, None))
        L.check(L.lib().adas_preprocess_ufld(dc.ptr, S, 720, 1280, lt.ptr, 320, 1600, C.c_double(0.6), None))
        L.check(L.lib().adas_synchronize())   # the stand-alone kernels ran on the null stream; the pipeline's streams are non-blocking
        pb.step(dt.ptr, lt.ptr); pb.sync()
        for s in range(S):
            a, b = PP.YoloPost.fetch(pa.post, s), PP.YoloPost.fetch(pb.post, s)
            for key in ("cand_anchor", "cand_conf", "keep", "xyxy_int"):
                np.testing.assert_array_equal(a[key], b[key])
            assert pa.decode.fetch(s) == pb.decode.fetch(s)
        dc.free()
    pa.close(); pb.close(); dt.free(); lt.free()


def test_pipeline_with_ufld_v1_lane_model():
    """The fused step with a UFLD v1 lane model (one output tensor, v1 decoder): lane results equal the stand-alone path."""
    from oracle import ufld_decode as UD
    S = 2
    lane_path, _, _ = netutil.model("ufld_v1_res18")
    c = UD.ModelConfigV1("tusimple")
    cfg = dict(griding_num=c.griding_num, cls_num_per_lane=c.cls_num_per_lane, img_w=c.img_w, img_h=c.img_h, row_anchor=c.row_anchor)
    pipe = PL.AdasPipeline(None, lane_path, n_streams=S, precision="bf16", src_hw=(720, 1280), use_graph=True, lane_cfg=cfg, track=False)
    x = netutil.lane_frames(S, 288, 800, seed=5)
    dx = L.DeviceBuffer.from_array(x)
    pipe.step(None, dx.ptr); pipe.sync()
    eng = CE.HipEngine(lane_path, "bf16", S)
    dec = PP.Ufld1Decode(c.griding_num, c.cls_num_per_lane, c.img_w, c.img_h, 800, 288, 1280, 720, c.row_anchor, S)
    want = dec.run_host(eng.engine_inference(x)[0])
    for s in range(S):
        assert pipe.decode.fetch(s) == want[s]
    assert sum(len(l) for l in want[0][0]) > 20
    pipe.close(); eng.close(); dec.close(); dx.free()


@pytest.mark.parametrize("split,frames_per_step,streams", [
    (["--streams", "2"], 4, [2, 2]),            # weak scaling: every rank its own 2 streams
    (["--total-streams", "3"], 3, [2, 1]),      # a 3-stream JOB dealt by sharding.streams_of_rank: stream s -> rank s mod 2
    (["--total-streams", "1"], 1, [1, 0]),      # fewer streams than ranks: rank 1 owns nothing and only takes part in the collectives
])
def test_bench_gpus_n_relaunches_n_ranks(tmp_path, split, frames_per_step, streams):
    """`python bench.py --gpus 2` with no torchrun environment must re-execute itself under torch.distributed.run and print ONE line
    with n_gpus = 2 and two per-rank records.  Rehearsed on this 1-GPU box with both ranks sharing the device and gloo carrying
    the statistics (ADAS_BENCH_SHARE_GPU / ADAS_BENCH_BACKEND exist for exactly this); the driver's 8-GPU run takes the same path
    with RCCL.  `--total-streams N` runs the stream router's uneven splits (13-over-8 in tests/test_multigpu_gloo.py) for real."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, ADAS_BENCH_BACKEND="gloo", ADAS_BENCH_SHARE_GPU="1", ADAS_MODEL_DIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", *split,
                          "--no-cpu-baseline", "--precision", "fp16", "--repeats", "1", "--latency-steps", "8"], env=env, capture_output=True,
                         text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and len(r["per_rank"]) == 2 and r["config"]["frames_per_step"] == frames_per_step
    assert [q["streams"] for q in r["per_rank"]] == streams
    assert r["scaling"] == ("strong" if split[0] == "--total-streams" else "weak")
    assert r["value"] > 0 and r["config"]["frames_at_candidate_capacity"] == 0
    # whole-job rate = every rank's frames / the slowest rank's seconds (the max-over-ranks clock), not a sum of per-rank rates
    frames, slowest = sum(q["frames"] for q in r["per_rank"]), max(q["seconds"] for q in r["per_rank"])
    assert frames == 3 * frames_per_step and abs(r["value"] - frames / slowest) <= 0.02 * r["value"], (r["value"], r["per_rank"])
    assert r["cpu_baseline"] is None and r["modes"] is None          # single-rank legs are skipped with world > 1
    # every rank reports where its launching thread was pinned (N > 1 only): a share of the CPUs, or its GPU's NUMA node
    assert all(q["cpus"] >= 1 and isinstance(q["pinned"], bool) for q in r["per_rank"])


def test_pipeline_step_frames_host_equals_device_frames(tmp_path):
    """adas_pipeline_step_frames_host (pinned host frames, copy stream, two staging buffers) gives exactly the results of
    adas_pipeline_step_frames on device-resident frames, over enough steps to cycle both staging buffers."""
    import bench
    S = 2
    cam = [bench.cam_frames(S, 170 + i) for i in range(3)]
    lane_path, _, _ = netutil.model("ufldv2_res18", **LANE_KW)
    det_path = M.build("yolov8n").save(str(tmp_path / "d.hipm"))
    pa = PL.AdasPipeline(det_path, lane_path, n_streams=S, src_hw=(720, 1280), use_graph=True, lane_cfg=LANE_CFG)
    pb = PL.AdasPipeline(det_path, lane_path, n_streams=S, src_hw=(720, 1280), use_graph=True, lane_cfg=LANE_CFG)
    pinned = [L.PinnedBuffer(c.shape) for c in cam]
    for b, c in zip(pinned, cam):
        b.array[...] = c
    dev = [L.DeviceBuffer.from_array(c) for c in cam]
    one = L.PinnedBuffer(cam[0].shape)                 # a single host buffer refilled for every step: wait_upload() before each refill
    for i, k in enumerate((0, 1, 2, 0, 2, 1, 1)):
        if i % 2:
            pa.wait_upload()
            one.array[...] = cam[k]
            pa.step_frames_host(one.ptr, (720, 1280), 0.6)
        else:
            pa.step_frames_host(pinned[k].ptr, (720, 1280), 0.6)
        pb.step_frames(dev[k].ptr, (720, 1280), 0.6)
    pa.sync(); pb.sync()
    for s in range(S):
        a, b = PP.YoloPost.fetch(pa.post, s), PP.YoloPost.fetch(pb.post, s)
        for key in ("cand_anchor", "cand_conf", "keep", "xyxy_int"):
            np.testing.assert_array_equal(a[key], b[key])
        assert pa.decode.fetch(s) == pb.decode.fetch(s)
        pc.check_track_frame(gpu_api.track_snapshot(*pa.tracker.fetch(s)), gpu_api.track_snapshot(*pb.tracker.fetch(s)), ctx=s)
    pa.close(); pb.close()
    for b in pinned + [one]:
        b.free()
    for d in dev:
        d.free()


def test_changed_crop_ratio_or_config_recaptures_the_graph(tmp_path):
    """A captured step bakes the lane crop ratio and the decoder configuration into its kernel arguments: the same frame buffer
    with another crop ratio, or after a configuration setter, must not replay the stale capture (round-1 advisor finding)."""
    import bench
    S = 1
    cam = bench.cam_frames(S, 222)
    lane_path, _, _ = netutil.model("ufldv2_res18", **LANE_KW)
    pg = PL.AdasPipeline(None, lane_path, n_streams=S, src_hw=(720, 1280), use_graph=True, lane_cfg=LANE_CFG, track=False)
    pe = PL.AdasPipeline(None, lane_path, n_streams=S, src_hw=(720, 1280), use_graph=False, lane_cfg=LANE_CFG, track=False)
    dc = L.DeviceBuffer.from_array(cam)
    seen = []
    for crop in (0.6, 0.8, 0.6):
        pg.step_frames(dc.ptr, (720, 1280), crop); pg.sync()
        pe.step_frames(dc.ptr, (720, 1280), crop); pe.sync()
        assert pg.decode.fetch(0) == pe.decode.fetch(0), crop
        seen.append(pg.decode.fetch(0))
    assert seen[0] == seen[2]
    pg.close(); pe.close(); dc.free()


@pytest.mark.parametrize("det_name,precision", [("yolov8n", "fp16"), ("yolov7-tiny", "fp16"), ("yolov8n", "fp16x3")])
def test_detect_sink_off_gives_the_same_step(tmp_path, monkeypatch, det_name, precision):
    """ADAS_NO_DETECT_SINK=1 (full head + class scan inside the step) against the default (per-anchor maxima straight from the Detect
    kernel): identical candidates, survivors and tracks; a detector engine keeps returning the whole head to engine_inference callers.
    fp16x3: the exact mode's fused Detect (detect_v8_fused_x3_kernel) and its sink."""
    import bench
    S = 2
    cam = [bench.cam_frames(S, 90 + i) for i in range(2)]
    seam = np.concatenate([importlib.import_module("oracle.preprocess").yolo_prepare_input(f, (640, 640)) for f in cam[0]])
    det_path, _, _ = bench.build_detector(M, CE, det_name, seam, str(tmp_path), "sink", target_per_frame=60.0)   # v8 layout / v5 layout
    lane_path, _, _ = netutil.model("ufldv2_res18")
    kw = dict(n_streams=S, precision=precision, src_hw=(720, 1280), use_graph=True, head_layout=L.HEAD_V8 if det_name == "yolov8n" else L.HEAD_V5)
    pa = PL.AdasPipeline(det_path, lane_path, **kw)
    monkeypatch.setenv("ADAS_NO_DETECT_SINK", "1")
    pb = PL.AdasPipeline(det_path, lane_path, **kw)
    monkeypatch.delenv("ADAS_NO_DETECT_SINK")
    assert L.lib().adas_pipeline_detect_sink(pa.h) == 1 and L.lib().adas_pipeline_detect_sink(pb.h) == 0
    n = 0
    for k in (0, 1, 1, 0):
        dc = L.DeviceBuffer.from_array(cam[k])
        pa.step_frames(dc.ptr, (720, 1280), 0.6); pa.sync()
        pb.step_frames(dc.ptr, (720, 1280), 0.6); pb.sync()
        for s in range(S):
            a, b = PP.YoloPost.fetch(pa.post, s), PP.YoloPost.fetch(pb.post, s)
            for key in ("cand_anchor", "cand_conf", "cand_cls", "cand_xywh", "keep", "xywh", "conf", "class_id", "xyxy_int"):
                np.testing.assert_array_equal(a[key], b[key], err_msg=key)
            n += len(a["cand_anchor"])
            pc.check_track_frame(gpu_api.track_snapshot(*pa.tracker.fetch(s)), gpu_api.track_snapshot(*pb.tracker.fetch(s)), ctx=(k, s))
        dc.free()
    assert n > 50
    # the engine inside the sink pipeline still hands the whole head to a host caller (the sink is set only around the step's launches)
    x = seam[:S].astype(np.float32)
    ha, hb = pa.det.engine_inference(x)[0], pb.det.engine_inference(x)[0]
    np.testing.assert_array_equal(ha, hb)
    assert float(np.abs(ha[:, 4:] if det_name == "yolov8n" else ha[..., 5:]).max()) > 0.0
    pa.close(); pb.close()


def test_detect_sink_needs_a_post_processor_of_the_same_head(tmp_path):
    """A v5-layout detector behind a post-processor created for the v8 reading of its (1, 25200, 85) head (legal: the shapes are
    consistent) must not get the sink: the engine would write 25,200 entries per frame into scan arrays sized for 85 anchors."""
    det_path = M.build("yolov7-tiny").save(str(tmp_path / "v7.hipm"))
    p = PL.AdasPipeline(det_path, None, n_streams=2, precision="fp16", src_hw=(720, 1280), use_graph=False)     # head_layout defaults to v8
    assert L.lib().adas_pipeline_detect_sink(p.h) == 0
    p.close()
    p = PL.AdasPipeline(det_path, None, n_streams=2, precision="fp16", src_hw=(720, 1280), use_graph=False, head_layout=L.HEAD_V5)
    assert L.lib().adas_pipeline_detect_sink(p.h) == 1
    p.close()
