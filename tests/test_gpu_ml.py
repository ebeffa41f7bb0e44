"""GPU: the multi-layer persistent launch (csrc/conv_ml.hip) against the per-layer launches it replaces.  The tile bodies are the same
code (conv_halo_body.h) or the same arithmetic in the same order (the pointwise tile), so EVERYTHING is compared bit for bit: head
outputs and every member layer's activation, over several forwards with fresh inputs (a stale read of another workgroup's data -- L1 or
cross-XCD -- shows up as a differing word), eager and under hipGraph replay with the lane network on the pipeline's second stream."""
import importlib
import os

import numpy as np
import pytest

import netutil

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def CE():
    from conftest import load_pkg
    load_pkg()
    ce = importlib.import_module("adas_amd.coreEngine")
    assert ce.L.lib().adas_device_count() > 0
    return ce


def _engine(CE, path, prec, batch, ml, **env):
    """An engine with the multi-layer launches on / off (the switch is read from the environment when the engine is created)."""
    old = {k: os.environ.get(k) for k in ("ADAS_ML",) + tuple(env)}
    os.environ["ADAS_ML"] = "1" if ml else "0"
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        e = CE.HipEngine(path, precision=prec, max_batch=batch)
        e.prepare(batch)          # (the tables read ADAS_ML_* too)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return e


def _frames(n, seed):
    rng = np.random.default_rng(seed)
    x = netutil.coco_like_frames(2)
    reps = [np.roll(x[i % 2], (int(rng.integers(0, 640)), int(rng.integers(0, 640))), (1, 2)) * np.float32(rng.uniform(0.6, 1.0)) for i in range(n)]
    return np.ascontiguousarray(np.stack(reps)).astype(np.float32)


@pytest.mark.parametrize("name,prec,batch", [("yolov8n", "fp16", 64), ("yolov8n", "bf16", 16), ("yolov8s", "fp16", 16), ("yolov8n", "fp16", 3)])
def test_multi_layer_launch_is_bit_identical_to_per_layer_launches(CE, name, prec, batch):
    path, W, g = netutil.model(name)
    a = _engine(CE, path, prec, batch, ml=True)
    b = _engine(CE, path, prec, batch, ml=False)
    info = a.ml_info(batch)
    print(name, prec, batch, "multi-layer launches:", info, "kernel launches per forward:", a.launch_count(batch), "vs", b.launch_count(batch))
    assert b.ml_info(batch) == {"launches": 0, "layers": 0, "items": 0}
    if batch == 64 and name == "yolov8n":
        assert info["launches"] >= 3 and info["layers"] >= 30, info                      # the 40x40 / 20x20 layers and the Detect branches
        assert a.launch_count(batch) <= b.launch_count(batch) - 25
    assert info["launches"] >= 1 and info["layers"] >= 2 * info["launches"], info         # (smaller batches pack narrower channel blocks: fewer eligible layers)
    n_layers = a.stats()["num_layers"]
    members = [i for i in range(n_layers) if a.layer_kernel(i, batch).startswith(("conv_ml_kernel", "(in the multi-layer"))]
    assert len(members) == info["layers"]
    for rep in range(3):
        x = _frames(batch, 100 + rep)
        ya, yb = a.engine_inference(x), b.engine_inference(x)
        assert a.ml_status(batch) == 0
        for u, v in zip(ya, yb):
            assert np.array_equal(np.asarray(u), np.asarray(v)), "head differs (rep %d)" % rep
        if rep == 2:
            for i in members:
                u, v = a.fetch_activation(i, batch), b.fetch_activation(i, batch)
                assert np.array_equal(u, v), "layer %d (%s) differs" % (i, a.layer_info(i)[0])
    a.close(); b.close()


@pytest.mark.parametrize("name,prec,batch", [("yolov8n", "fp16", 64), ("yolov8n", "bf16", 5), ("yolov8n", "fp16", 1), ("yolov8s", "fp16", 8), ("yolov8l", "fp16", 1)])
def test_grouped_launch_of_independent_layers_is_bit_identical(CE, name, prec, batch):
    """The default path: 3x3 halo convs of one dependency level of a run (the Detect branches of every pyramid level) share a launch
    (conv_halo_group_kernel).  Against ADAS_NO_GROUP=1 -- every layer its own conv_halo launch: heads and member activations bit for bit,
    fewer launches."""
    path, W, g = netutil.model(name)
    a = _engine(CE, path, prec, batch, ml=False)
    b = _engine(CE, path, prec, batch, ml=False, ADAS_NO_GROUP=1)
    n_layers = a.stats()["num_layers"]
    ka = [a.layer_kernel(i, batch) for i in range(n_layers)]
    kb = [b.layer_kernel(i, batch) for i in range(n_layers)]
    members = [i for i in range(n_layers) if ka[i].startswith(("conv_halo_group_kernel", "(in the grouped launch"))]
    print(name, prec, batch, "grouped layers:", len(members), "launches", a.launch_count(batch), "vs", b.launch_count(batch), [k for k in ka if k.startswith("conv_halo_group")])
    assert not any(k.startswith(("conv_halo_group_kernel", "(in the grouped")) for k in kb)
    assert all(kb[i].startswith("conv_halo_kernel") for i in members)
    if name == "yolov8n" and batch == 64:
        assert len(members) == 10 and a.launch_count(batch) == b.launch_count(batch) - 8      # Detect: ten launches become two
    if batch == 1:      # one frame at a time: all twelve Detect convs (narrow channel blocks, no conv_halo_rw at this size) in two launches
        assert len(members) >= 12 and a.launch_count(batch) <= b.launch_count(batch) - 10, (len(members), a.launch_count(batch), b.launch_count(batch))
    for rep in range(2):
        x = _frames(batch, 300 + rep)
        ya, yb = a.engine_inference(x), b.engine_inference(x)
        for u, v in zip(ya, yb):
            assert np.array_equal(np.asarray(u), np.asarray(v)), "head differs (rep %d)" % rep
    for i in members:
        assert np.array_equal(a.fetch_activation(i, batch), b.fetch_activation(i, batch)), (i, a.layer_info(i)[0])
    a.close(); b.close()


def test_layer_major_order_and_small_grid_give_the_same_bits(CE):
    """The ticket order and the number of resident workgroups are scheduling only: layer-major tickets and a 96-workgroup grid (every
    dependency wait exposed) produce the bits of the default launch."""
    path, W, g = netutil.model("yolov8n")
    ref = _engine(CE, path, "fp16", 16, ml=False)
    x = _frames(16, 7)
    want = [np.array(v, copy=True) for v in ref.engine_inference(x)]
    ref.close()
    for env in (dict(ADAS_ML_ORDER=0), dict(ADAS_ML_GRID=96), dict(ADAS_ML_ORDER=0, ADAS_ML_GRID=40)):
        e = _engine(CE, path, "fp16", 16, ml=True, **env)
        got = e.engine_inference(x)
        assert e.ml_status(16) == 0
        for u, v in zip(got, want):
            assert np.array_equal(np.asarray(u), v), env
        e.close()


def test_dependency_wait_is_bounded(CE):
    """Every wait in the kernel gives up after ADAS_ML_SPIN polls: with a limit of 1 poll and 16 workgroups chasing 30 dependent layers,
    waits DO time out -- the launch must come back (no hang), raise its error word and be reported by adas_engine_ml_status; a
    following launch with a sane limit on a fresh engine is clean."""
    path, W, g = netutil.model("yolov8n")
    e = _engine(CE, path, "fp16", 16, ml=True, ADAS_ML_SPIN=1, ADAS_ML_ORDER=0)
    x = _frames(16, 9)
    try:
        e.engine_inference(x)                  # returns: bounded -- with an error when a wait gave up (the items behind it were skipped,
        word = e.ml_status(16)                 # not computed on incomplete inputs: the caller never sees partial outputs)
    except Exception as ex:
        word = -1
        assert "timed out" in str(ex)
    print("spin limit 1: status", word)
    e.close()
    e = _engine(CE, path, "fp16", 16, ml=True)
    e.engine_inference(x)
    assert e.ml_status(16) == 0
    e.close()


def test_pipeline_with_multi_layer_launches_matches_per_layer_pipeline(tmp_path):
    """The whole captured step (hipGraph replay, lane network on the second stream, pre-processing, NMS, tracker) with the detector's
    multi-layer launches against the same step with per-layer launches: candidates, survivors, boxes, tracks identical over 6 steps."""
    from conftest import load_pkg
    load_pkg()
    import bench
    import gpu_api
    L = importlib.import_module("adas_amd._lib")
    M = importlib.import_module("adas_amd.models")
    PL = importlib.import_module("adas_amd.pipeline")
    PP = importlib.import_module("adas_amd.postproc")
    S = 8
    cams = [bench.cam_frames(S, 900 + i) for i in range(3)]
    lane_path, _, _ = netutil.model("ufldv2_res18")
    det_path = M.build("yolov8n").save(str(tmp_path / "d.hipm"))
    pipes = []
    for ml in (True, False):
        old = os.environ.get("ADAS_ML")
        os.environ["ADAS_ML"] = "1" if ml else "0"
        try:
            pipes.append(PL.AdasPipeline(det_path, lane_path, n_streams=S, precision="fp16", src_hw=(720, 1280), use_graph=True))
        finally:
            if old is None:
                os.environ.pop("ADAS_ML", None)
            else:
                os.environ["ADAS_ML"] = old
    dev = [L.DeviceBuffer.from_array(c) for c in cams]
    for k in (0, 1, 2, 0, 1, 2):
        for p in pipes:
            p.step_frames(dev[k].ptr, (720, 1280), 0.6)
        for p in pipes:
            p.sync()
        for s in range(S):
            a, b = PP.YoloPost.fetch(pipes[0].post, s), PP.YoloPost.fetch(pipes[1].post, s)
            for key in ("cand_anchor", "cand_conf", "keep", "xyxy_int"):
                assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), (k, s, key)
            assert gpu_api.track_snapshot(*pipes[0].tracker.fetch(s)) == gpu_api.track_snapshot(*pipes[1].tracker.fetch(s)), (k, s)
            assert pipes[0].decode.fetch(s) == pipes[1].decode.fetch(s)
    assert pipes[0].det.ml_info(S)["launches"] >= 3 and pipes[1].det.ml_info(S)["launches"] == 0
    assert pipes[0].det.ml_status(S) == 0
    for p in pipes:
        p.close()
    for b in dev:
        b.free()
