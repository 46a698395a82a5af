#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace directory, in dispatch order, folded by position in a
repeating pattern: `trace_kernel_order.py <dir> <kernel substring> <period>` prints the mean duration at each position of the
period (e.g. period 2 for the two LayerNorm GEMMs of an encoder layer: attention output, then FFN down)."""
import csv, glob, sys, collections

d, sub, period = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if sub in r["Kernel_Name"]]
by = collections.defaultdict(list)
for i, x in enumerate(dur):
    by[i % period].append(x)
print(sub, "dispatches", len(dur))
for k in sorted(by):
    v = sorted(by[k])
    print(f"  position {k}: n {len(v)}  median {v[len(v)//2]/1e3:.2f} us  min {v[0]/1e3:.2f}  max {v[-1]/1e3:.2f}")
