"""CPU, world_size 2 over gloo: the stream-sharding + statistics path bench.py uses for --gpus N.
No data-path collective exists; what is tested is placement, the max-over-ranks clock and the stats all_gather."""
import importlib, os, socket, sys
import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_pkg, ROOT

load_pkg()
SH = importlib.import_module("adas_amd.sharding")


def test_assign_streams_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 513):
        for w in (1, 2, 4, 8):
            parts = SH.assign_streams(n, w)
            flat = sorted(s for p in parts for s in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            assert all(s % w == r for r, p in enumerate(parts) for s in p)
    with pytest.raises(ValueError):
        SH.assign_streams(4, 0)
    assert SH.aggregate_throughput([{"frames": 100, "seconds": 2.0}, {"frames": 100, "seconds": 4.0}])["fps"] == 50.0


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    sh = importlib.import_module("adas_amd.sharding")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    env = sh.RankEnv.from_environ()
    dist = sh.init_process_group(env, backend="gloo")
    mine = sh.streams_of_rank(9, env)
    elapsed = sh.max_over_ranks(1.0 + rank, dist)                       # rank 1 is the slow one
    # the per-rank record bench.py gathers: {frames, seconds, p50, p99} (+ the stream count for this test)
    stats = sh.gather_stats({"frames": 40.0 * len(mine), "seconds": 1.0 + rank, "streams": float(len(mine)), "p50_ms": 5.0 + rank, "p99_ms": 5.5 + rank},
                            ("frames", "seconds", "streams", "p50_ms", "p99_ms"), dist)
    dist.barrier()
    q.put((rank, mine, elapsed, stats))
    dist.destroy_process_group()


def test_two_rank_gloo_stats_gather():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, s0, e0, st0), (r1, s1, e1, st1) = res
    assert s0 == [0, 2, 4, 6, 8] and s1 == [1, 3, 5, 7]
    assert e0 == e1 == 2.0                                                # max over ranks, identical everywhere
    assert st0 == st1 and [d["streams"] for d in st0] == [5.0, 4.0]
    assert [d["p50_ms"] for d in st0] == [5.0, 6.0] and [d["p99_ms"] for d in st0] == [5.5, 6.5]
    agg = SH.aggregate_throughput(st0)
    assert agg["frames"] == 360.0 and agg["seconds"] == 2.0 and agg["fps"] == 180.0


def _worker_n(rank, world, port, n_streams, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    sh = importlib.import_module("adas_amd.sharding")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    env = sh.RankEnv.from_environ()
    dist = sh.init_process_group(env, backend="gloo")
    mine = sh.streams_of_rank(n_streams, env)
    seconds = 1.0 + 0.25 * ((rank * 5) % world)                            # a different slowest rank per world size
    elapsed = sh.max_over_ranks(seconds, dist)
    stats = sh.gather_stats({"frames": 20.0 * len(mine), "seconds": seconds, "streams": float(len(mine)), "p50_ms": 4.0 + rank, "p99_ms": 4.5 + rank},
                            ("frames", "seconds", "streams", "p50_ms", "p99_ms"), dist)
    dist.barrier()
    q.put((rank, mine, elapsed, stats))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_streams", [(4, 9), (4, 2), (8, 13), (8, 64)])
def test_four_and_eight_rank_gloo_uneven_streams(world, n_streams):
    """The driver's 1 / 2 / 4 / 8 scaling runs rehearsed on CPU: world sizes 4 and 8 over gloo, stream counts that do not
    divide (9 over 4, 13 over 8) and one smaller than the world (2 over 4: two ranks own nothing and still take part in both
    collectives).  Whole-job frames/s = all frames / the slowest rank's seconds, identical on every rank."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, n_streams, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=240) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    owned = sorted(s_ for _, mine, _, _ in res for s_ in mine)
    assert owned == list(range(n_streams))                                                    # every stream on exactly one rank
    assert all(mine == list(range(r, n_streams, world)) for r, mine, _, _ in res)
    slowest = max(1.0 + 0.25 * ((r * 5) % world) for r in range(world))
    assert all(e == slowest for _, _, e, _ in res)
    st0 = res[0][3]
    assert all(st == st0 for _, _, _, st in res) and len(st0) == world
    assert [d["streams"] for d in st0] == [float(len(range(r, n_streams, world))) for r in range(world)]
    agg = SH.aggregate_throughput(st0)
    assert agg["frames"] == 20.0 * n_streams and agg["seconds"] == slowest and agg["fps"] == 20.0 * n_streams / slowest


def test_rank_affinity_helpers(tmp_path):
    """sharding.pin_rank's pieces without touching this process's affinity: sysfs cpulist parsing, the GPU's NUMA node from its PCI
    address, and the CPU set a rank's launching thread gets (the node's CPUs when known, else a contiguous share)."""
    assert SH.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and SH.parse_cpulist("") == []
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    assert SH.gpu_numa_node("0000:C1:00.0", str(tmp_path)) == 1
    assert SH.gpu_numa_node("0000:05:00.0", str(tmp_path)) == -1 and SH.gpu_numa_node(None) == -1
    allowed = list(range(16))
    assert SH.rank_cpus(3, 8, allowed, node_cpus=[8, 9, 10, 11, 40]) == [8, 9, 10, 11]       # the node's CPUs, within the allowed set
    assert SH.rank_cpus(3, 8, allowed, node_cpus=[40, 41]) == [6, 7]                        # node outside the cpuset: fall back to a share
    shares = [SH.rank_cpus(r, 8, allowed) for r in range(8)]
    assert shares == [[2 * r, 2 * r + 1] for r in range(8)]                                 # disjoint, contiguous, covering
    assert SH.rank_cpus(0, 1, allowed) == allowed and SH.rank_cpus(5, 8, [0, 1, 2]) == [0, 1, 2]
    # an idle rank (no stream of the job) reports 0 frames in 0 seconds: the aggregate ignores it in the per-rank rates
    agg = SH.aggregate_throughput([{"frames": 30.0, "seconds": 1.5}, {"frames": 0.0, "seconds": 0.0}])
    assert agg["fps"] == 20.0 and agg["min_rank_fps"] == 20.0 and agg["max_rank_fps"] == 20.0


def test_pipeline_for_rank_uses_the_stream_router(monkeypatch):
    """AdasPipeline.for_rank(total_streams) builds a rank's pipeline for exactly the streams sharding.streams_of_rank deals it (no GPU
    here: the constructor is replaced by a recorder) and returns None for a rank that owns nothing."""
    PL = importlib.import_module("adas_amd.pipeline")
    made = []

    def fake_init(self, det_model=None, lane_model=None, n_streams=1, **kw):
        made.append((det_model, lane_model, n_streams, kw))
        self.S, self.stream_ids, self.h = n_streams, list(range(n_streams)), None
    monkeypatch.setattr(PL.AdasPipeline, "__init__", fake_init)
    monkeypatch.setattr(PL.AdasPipeline, "close", lambda self: None)
    monkeypatch.setattr(PL.AdasPipeline, "__del__", lambda self: None)
    p = PL.AdasPipeline.for_rank("d.hipm", "l.hipm", 13, env=SH.RankEnv(rank=5, local_rank=5, world=8), precision="fp16x3")
    assert p.stream_ids == [5] and made[-1][2] == 1 and made[-1][3] == {"precision": "fp16x3"}
    p = PL.AdasPipeline.for_rank("d.hipm", "l.hipm", 13, env=SH.RankEnv(rank=2, local_rank=2, world=8))
    assert p.stream_ids == [2, 10] and made[-1][2] == 2
    assert PL.AdasPipeline.for_rank("d.hipm", "l.hipm", 2, env=SH.RankEnv(rank=3, local_rank=3, world=4)) is None
