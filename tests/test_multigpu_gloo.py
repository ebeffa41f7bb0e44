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
