#!/usr/bin/env python3
"""How much do the two branches of the step really overlap?  python tools/trace_overlap.py <rocprofv3 -d dir> [last_n_kernels]
Reads *kernel_trace.csv (rocprofv3 --kernel-trace): over the last N dispatches (the steady-state steps) prints wall time, the sum of
kernel durations, the time with >= 1 / >= 2 kernels in flight, and the same per queue."""
import csv, glob, os, sys, collections
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
rows = rows[-last:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, q, k in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy1 = busy2 = 0; depth = 0; prev = ev[0][0]
for t, dlt in ev:
    if depth >= 1: busy1 += t - prev
    if depth >= 2: busy2 += t - prev
    depth += dlt; prev = t
tot = sum(e - s for s, e, q, k in rows)
print("dispatches %d  wall %.3f ms  sum of kernel durations %.3f ms  >=1 in flight %.3f ms  >=2 in flight %.3f ms (%.1f %% of wall)" % (
    len(rows), (t1 - t0) / 1e6, tot / 1e6, busy1 / 1e6, busy2 / 1e6, 100.0 * busy2 / (t1 - t0)))
perq = collections.defaultdict(float)
for s, e, q, k in rows: perq[q] += e - s
for q, v in sorted(perq.items(), key=lambda kv: -kv[1]): print("  queue %s: %.3f ms of kernels" % (q, v / 1e6))
# which kernels run concurrently with conv_h8?
h8 = [(s, e) for s, e, q, k in rows if "conv_h8_kernel" in k]
other = collections.defaultdict(float)
for s, e, q, k in rows:
    if "conv_h8_kernel" in k: continue
    for hs, he in h8:
        ov = min(e, he) - max(s, hs)
        if ov > 0: other[k] += ov
h8tot = sum(e - s for s, e in h8)
print("conv_h8 in flight %.3f ms; other kernels overlapping it: %.3f ms total" % (h8tot / 1e6, sum(other.values()) / 1e6))
for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:8]: print("   %-60s %.3f ms" % (k, v / 1e6))
