#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv output per kernel:  python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if pat not in k:
            continue
        k = k.split("(")[0][-70:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add((f, r.get("Dispatch_Id")))
for k, c in agg.items():
    n = max(1, len(cnt[k]))
    print(f"== {k}  (dispatch-samples {n})")
    for name in sorted(c):
        print(f"   {name:32s} {c[name]:.4g}")
