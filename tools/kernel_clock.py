#!/usr/bin/env python3
"""Shader clock per kernel from a rocprofv3 `--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace` directory:
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so clock = counter / 8 / duration.  `kernel_clock.py <dir> [name filter]`"""
import csv, glob, sys, collections

d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(list)
for k, c in cnt.items():
    if flt in c["name"] and k in dur and dur[k] > 0:
        agg[c["name"][:60]].append((c.get("GRBM_GUI_ACTIVE", 0) / 8 / dur[k], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), dur[k]))
for name, v in agg.items():
    v.sort(key=lambda x: x[2])
    for lab, part in (("shorter half", v[: len(v) // 2]), ("longer half", v[len(v) // 2:])):
        if part:
            n = len(part)
            print(f"{name:60s} {lab:12s} n {n:4d}  {sum(x[2] for x in part)/n/1e3:8.2f} us  clock {sum(x[0] for x in part)/n:.3f} GHz  mfma busy {sum(x[1] for x in part)/n:.4g}")
