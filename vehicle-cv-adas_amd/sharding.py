"""Multi-GPU batched-video mode: independent video streams are the unit of parallelism.

Each stream owns its frames, its tracker state and (per GPU) a replica of the weights, so stream s simply lives on
rank s mod world: one process per GPU (torchrun), NO data-path collective.  RCCL (torch.distributed backend "nccl";
"gloo" in the CPU tests) is used only to agree on the wall-clock of a timed region and to gather per-rank throughput
statistics -- a few dozen bytes per rank, so xGMI bandwidth is irrelevant by construction.
"""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence


@dataclass
class RankEnv:
    rank: int
    local_rank: int
    world: int

    @classmethod
    def from_environ(cls, env=None):
        env = os.environ if env is None else env
        return cls(int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1")))


def assign_streams(n_streams: int, world: int) -> List[List[int]]:
    """Round-robin placement stream s -> rank s mod world (every rank gets floor or ceil of n/world streams)."""
    if world <= 0 or n_streams < 0:
        raise ValueError("world must be > 0 and n_streams >= 0")
    return [list(range(r, n_streams, world)) for r in range(world)]


def streams_of_rank(n_streams: int, env: RankEnv) -> List[int]:
    return assign_streams(n_streams, env.world)[env.rank]


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_node(pci_bdf: Optional[str], sysfs: str = "/sys") -> int:
    """NUMA node of a PCI device ('0000:c1:00.0'), -1 when the platform does not say."""
    if not pci_bdf:
        return -1
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bdf.lower(), "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def rank_cpus(local_rank: int, local_world: int, allowed: Sequence[int], node_cpus: Optional[Sequence[int]] = None) -> List[int]:
    """CPUs a rank's launching thread may run on.  With the GPU's NUMA node known: the node's CPUs (within the allowed set), divided
    among the ranks that share that node is left to the scheduler -- the point is locality to the GPU's root complex.  Without: a
    contiguous 1/local_world share of the allowed set, so eight launching threads do not migrate over each other (the frame-at-a-time
    path is bound by the launching thread, profiles/r05/b1_overlap.txt)."""
    allowed = sorted(allowed)
    if node_cpus:
        near = [c for c in allowed if c in set(node_cpus)]
        if near:
            return near
    if local_world <= 1 or len(allowed) < local_world:
        return list(allowed)
    per = len(allowed) // local_world
    return allowed[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank: int, local_world: int, pci_bdf: Optional[str] = None, sysfs: str = "/sys") -> Dict[str, object]:
    """Pin the calling process (its launching thread and the threads it starts later) near its GPU; never raises.
    -> {'numa_node', 'cpus' (count), 'first_cpu', 'pinned'} for the per-rank report."""
    info: Dict[str, object] = {"numa_node": -1, "cpus": 0, "first_cpu": -1, "pinned": False}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        node = gpu_numa_node(pci_bdf, sysfs)
        node_cpus = None
        if node >= 0:
            try:
                with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
                    node_cpus = parse_cpulist(f.read())
            except (OSError, ValueError):
                node_cpus = None
        cpus = rank_cpus(local_rank, local_world, allowed, node_cpus)
        info.update(numa_node=node, cpus=len(cpus), first_cpu=cpus[0] if cpus else -1)
        if cpus and len(cpus) < len(allowed):
            os.sched_setaffinity(0, cpus)
            info["pinned"] = True
    except (AttributeError, OSError, ValueError):
        pass
    return info


def init_process_group(env: RankEnv, backend: Optional[str] = None, device=None):
    """One process per GPU.  Returns torch.distributed (initialised) or None when world == 1."""
    if env.world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {}
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=env.rank, world_size=env.world, **kw)
    return dist


def max_over_ranks(value: float, dist, device="cpu") -> float:
    """The timed region ends when the slowest rank does."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stats(local: Dict[str, float], keys: Sequence[str], dist, device="cpu") -> List[Dict[str, float]]:
    """all_gather of one small fp64 vector per rank -> list (by rank) of dicts.  The only collective of the job."""
    if dist is None:
        return [dict((k, float(local[k])) for k in keys)]
    import torch
    mine = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [dict(zip(keys, (float(v) for v in t.cpu()))) for t in out]


def aggregate_throughput(per_rank: List[Dict[str, float]]) -> Dict[str, float]:
    """Whole-job frames/s = all frames / slowest rank's seconds; per-rank rates kept for the scaling report."""
    frames = sum(r["frames"] for r in per_rank)
    seconds = max(r["seconds"] for r in per_rank)
    rates = [r["frames"] / r["seconds"] for r in per_rank if r["seconds"] > 0]      # a rank that owns no stream reports 0 frames in 0 s
    return {"frames": frames, "seconds": seconds, "fps": frames / seconds if seconds > 0 else 0.0,
            "min_rank_fps": min(rates) if rates else 0.0, "max_rank_fps": max(rates) if rates else 0.0}
