#!/usr/bin/env python3
"""Print the device timeline of the last passes of a rocprofv3 --kernel-trace --memory-copy-trace run:
start offset, duration and gap of every kernel / copy (tools/profile_round.sh, fixed-cost analysis)."""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
tail = ev[-int(sys.argv[2]) if len(sys.argv) > 2 else -16:]
t0 = tail[0][0]
prev = None
for s, e, n in tail:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {n}")
    prev = e
