#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in KiB per dispatch).
    python tools/traffic_summary.py <dir> <kernel-substring> <kernel-label> <streams> [precision] [commit] > traffic.json
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of 16 B/lane reads -> x2.
Cross-checked on this kernel's own access pattern: the single-layer run (profiles/r01/pmc_mem_halo64_80x400_b64.txt)
reads 262 MB of activations (x1.27 window halo) and reports FETCH_SIZE = 158.6 MB uncorrected, 317 MB corrected."""
import csv, glob, json, os, sys
d, pat, label, streams = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
precision = sys.argv[5] if len(sys.argv) > 5 else "fp16"
commit = sys.argv[6] if len(sys.argv) > 6 else None
import datetime
tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r.get("Kernel_Name", "") and r["Counter_Name"] in tot:
            t = tot[r["Counter_Name"]]
            t[0] += float(r["Counter_Value"]); t[1] += 1
fetch_kib = tot["FETCH_SIZE"][0] / max(1, tot["FETCH_SIZE"][1])
write_kib = tot["WRITE_SIZE"][0] / max(1, tot["WRITE_SIZE"][1])
hbm = 2.0 * fetch_kib * 1024 + write_kib * 1024
print(json.dumps({"kernel": label, "streams": streams, "precision": precision, "commit": commit,
                  "collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "hbm_bytes_per_launch": round(hbm),
                  "fetch_kib_raw": round(fetch_kib, 1), "write_kib_raw": round(write_kib, 1),
                  "launches_sampled": [tot["FETCH_SIZE"][1], tot["WRITE_SIZE"][1]],
                  "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-overlap --steps 5`, "
                            "average per dispatch of the named kernel; FETCH_SIZE x2 (gfx950 16 B/lane correction), WRITE_SIZE as reported"}))
