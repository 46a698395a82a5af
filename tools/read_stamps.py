#!/usr/bin/env python3
"""Where the waves of scan_mfma8_kernel spent their time (diagnostic build, tools/build_stamps.sh).
    python tools/read_stamps.py stamps.bin [pass index, default last]
Words per wave: 0 entry, 1 tile staged, 3 end of its last block (100 MHz clock); 4 blocks, 5 ticks inside the fine screen,
6 blocks that reached the fine screen, 7 blocks that passed the block test; 2: ticks of the blocks that ended at the block test
(low word) and of those with a survivor (high word), each block booked from the end of the one before it."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 65536, 8)
k = int(sys.argv[2]) if len(sys.argv) > 2 else -1
w = a[k]
w = w[w[:, 0] != 0]
t0 = w[:, 0].min()
us = lambda x: (x.astype(np.int64) - int(t0)) / 100.0
ent, staged, end = us(w[:, 0]), us(w[:, 1]), us(w[:, 3])
q = lambda x: " ".join(f"{v:8.1f}" for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))
print(f"pass {k} of {a.shape[0]}: {w.shape[0]} waves        min      p10      p50      p90      p99      max   (us from the first wave's entry)")
print("entry             ", q(ent))
print("tile staged       ", q(staged))
print("end of last block ", q(end))
print("stage duration    ", q(staged - ent))
print("stream duration   ", q(end - staged))
print("blocks per wave   ", q(w[:, 4].astype(float)))
print("fine screen us    ", q(w[:, 5].astype(float) / 100.0))
print("blocks w/ survivor", q(w[:, 6].astype(float)))
print("blocks past pretest", q(w[:, 7].astype(float)))
cold_t, hot_t = (w[:, 2] & 0xffffffff).astype(float) / 100.0, (w[:, 2] >> 32).astype(float) / 100.0
nhot = w[:, 7].astype(float)
ncold = w[:, 4].astype(float) - nhot
print("us per block that ended at the block test", q(cold_t / np.maximum(ncold, 1)))
print("us per block with a survivor              ", q(hot_t / np.maximum(nhot, 1)), " (fine screen included)")
print("  of which outside the fine screen        ", q((hot_t - w[:, 5].astype(float) / 100.0) / np.maximum(nhot, 1)))
# when do the slowest waves lose their time: stream duration against fine-screen time
d = end - staged
f = w[:, 5].astype(float) / 100.0
print("corr(stream duration, fine-screen time) =", float(np.corrcoef(d, f)[0, 1]))
print("mean stream duration minus fine screen  =", float((d - f).mean()), "us;  spread of that (p1..p99):", q(d - f))
# who is slow: by XCD (workgroups are dealt round-robin over the 8 XCDs), by wave of the workgroup, by workgroup of a CU
idx = np.nonzero(a[k][:, 0] != 0)[0]
wg, wv = idx // 4, idx % 4
for name, key in (("workgroup % 8 (XCD)", wg % 8), ("wave of the workgroup", wv), ("workgroup // 256 (round of the launch)", wg // 256)):
    print(name + ":", "  ".join(f"{g}: {d[key == g].mean():.0f}" for g in np.unique(key)))
per_wg = np.array([d[wg == g].mean() for g in np.unique(wg)])
print("workgroup means: p1..p99", q(per_wg), " spread inside a workgroup (mean of max - min):", float(np.mean([np.ptp(d[wg == g]) for g in np.unique(wg)])))
