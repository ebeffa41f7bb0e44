#!/usr/bin/env python3
"""Same-box A/B of engine switches: runs `bench.py --no-cpu-baseline --no-extras` once per variant (the switches are environment variables
read once per process, so every variant is its own process), interleaved over `--rounds` passes so that clock / thermal drift hits all
variants alike, and prints frames/s (median of the repeats of every pass) and the per-net stage times.

    python tools/ab_bench.py --variant base --variant nofuse:ADAS_NO_C2F_FUSE=1 --variant bm:ADAS_HALO_BM128=512 [--preset north-star] [--rounds 2]

Box-to-box spread on the gpurun pool is +-4 %, run-to-run on one box +-0.5 %: differences below ~1 % need --rounds >= 3."""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--variant", action="append", required=True, help="name[:ENV=VAL[,ENV=VAL...]]")
ap.add_argument("--preset", default="north-star")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--extra", default="", help="further bench.py arguments, space separated")
a = ap.parse_args()

variants = []
for v in a.variant:
    name, _, envs = v.partition(":")
    variants.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
res = {name: [] for name, _ in variants}
for r in range(a.rounds):
    for name, env in variants:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--preset", a.preset, "--no-cpu-baseline", "--no-extras", "--steps", str(a.steps),
               "--repeats", str(a.repeats)] + a.extra.split()
        p = subprocess.run(cmd, env=dict(os.environ, ADAS_BENCH_NO_PMC="1", **env), capture_output=True, text=True, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            print(f"{name}: bench failed (rc {p.returncode}): {p.stderr.strip().splitlines()[-1:]}" )
            continue
        d = json.loads(line[-1])
        fps = [d["value"]] + ([d["repeats"]["fps_median"]] if d.get("repeats") else [])
        res[name].append(dict(fps=sorted(fps)[len(fps) // 2], stages=d.get("stages", {})))
        print(f"round {r} {name:16s} {res[name][-1]['fps']:10.1f} frames/s  {d.get('stages')}", flush=True)
base = None
print("\nvariant            median fps   vs first   det_net_ms  lane_net_ms")
for name, _ in variants:
    rs = res[name]
    if not rs:
        continue
    fps = sorted(x["fps"] for x in rs)[len(rs) // 2]
    base = base or fps
    det = sorted(x["stages"].get("det_net_ms", 0) for x in rs)[len(rs) // 2]
    lane = sorted(x["stages"].get("lane_net_ms", 0) for x in rs)[len(rs) // 2]
    print(f"{name:16s} {fps:12.1f}   {100 * (fps / base - 1):+6.2f} %   {det:9.4f}   {lane:9.4f}")
