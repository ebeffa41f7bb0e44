#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc + --kernel-trace run:  python tools/pmc_top.py <dir> [top_n]
For every kernel (aggregated over its dispatches): launches, average duration from the kernel trace, the raw counters, and derived
figures -- MFMA pipe busy % (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)), waves waiting % (SQ_WAIT_ANY / SQ_WAVE_CYCLES), VALU per
MFMA, and the shader clock the launch ran at (cycles / wall duration: the DVFS clock under that load, MI355X_MICROARCH.md "DVFS
give-back").  rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs: cycles = GRBM_GUI_ACTIVE / 8."""
XCDS = 8
import csv, glob, os, sys, collections
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctr = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
dur = collections.defaultdict(list); name_of = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0]
        ctr[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r.get("Dispatch_Id"))
        name_of[r.get("Dispatch_Id")] = k
        s, e = r.get("Start_Timestamp"), r.get("End_Timestamp")
        if s and e and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append(float(e) - float(s))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0]
        if not dur.get(k):
            pass
        dur[k + "#trace"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = []
for k, c in ctr.items():
    n = max(1, len(disp[k]))
    t = dur.get(k) or dur.get(k + "#trace") or []
    avg_ns = sum(t) / len(t) if t else 0.0
    rows.append((c.get("GRBM_GUI_ACTIVE", 0.0), k, n, avg_ns, c))
rows.sort(reverse=True)
for gui, k, n, avg_ns, c in rows[:top]:
    g = c.get("GRBM_GUI_ACTIVE", 0.0) / n / XCDS      # cycles of the launch
    print("== %s" % k[-110:])
    print("   dispatches %d   avg duration %.1f us   cycles/launch %.4g (GRBM_GUI_ACTIVE / 8 XCDs)" % (n, avg_ns / 1e3, g))
    if g > 0:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print("   MFMA pipe busy           %.1f %%   (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs))" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (g * 1024)))
        if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
            print("   waves waiting            %.1f %%   (SQ_WAIT_ANY / SQ_WAVE_CYCLES)" % (100.0 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
        if c.get("SQ_INSTS_MFMA"):
            print("   VALU per MFMA            %.2f      MFMA instructions/launch %.4g" % (c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"], c["SQ_INSTS_MFMA"] / n))
        if avg_ns > 0:
            print("   shader clock             %.2f GHz  (cycles / wall duration)" % (g / avg_ns))
        if "SQ_WAVES" in c:
            print("   waves/launch             %.4g" % (c["SQ_WAVES"] / n))
    for name in sorted(c):
        print("   %-28s %.5g" % (name, c[name] / n))
