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
WPB = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # 12: the DRAIN form, wave 11 of a workgroup is the drain wave
w = a[k]
t0 = w[w[:, 0] != 0][:, 0].min()
us = lambda x: (x.astype(np.int64) - int(t0)) / 100.0
q = lambda x: " ".join(f"{v:8.1f}" for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))
if WPB == 12:
    allidx = np.nonzero(w[:, 0] != 0)[0]
    dr = w[allidx[allidx % 12 == 11]]
    print(f"drain waves: {dr.shape[0]}                        min      p10      p50      p90      p99      max")
    print("end (us from first entry)  ", q(us(dr[:, 3])))
    print("shader clock, MHz          ", q(dr[:, 1].astype(float) / np.maximum((dr[:, 3] - dr[:, 0]).astype(float), 1) * 100.0))
    print("entries out of the ring    ", q(dr[:, 4].astype(float)))
    print("shed by a repeated test    ", q(dr[:, 7].astype(float)))
    print("rounds of eight worked     ", q(dr[:, 6].astype(float)))
    print("us at work                 ", q(dr[:, 5].astype(float) / 100.0))
    print("us per round               ", q(dr[:, 5].astype(float) / 100.0 / np.maximum(dr[:, 6].astype(float), 1)))
    print("most waiting in the ring   ", q((dr[:, 2] >> 32).astype(float)))
    print("most waiting on the stack  ", q((dr[:, 2] & 0xffffffff).astype(float)))
    keepmask = np.ones(w.shape[0], bool)
    keepmask[11::12] = False
    w = np.where(keepmask[:, None], w, 0)
a = a.copy()
a[k] = w
w = w[w[:, 0] != 0]
ent, staged, end = us(w[:, 0]), us(w[:, 1]), us(w[:, 3])
print(f"pass {k} of {a.shape[0]}: {w.shape[0]} waves        min      p10      p50      p90      p99      max   (us from the first wave's entry)")
print("entry             ", q(ent))
print("tile staged       ", q(staged))
print("end of last block ", q(end))
print("stage duration    ", q(staged - ent))
print("stream duration   ", q(end - staged))
print("blocks per wave   ", q(w[:, 4].astype(float)))
if len(sys.argv) > 4 and sys.argv[4] == "timeline":  # PCV_STAMPS_TIMELINE build: slots 1, 5, 6, 7 = (shader clock << 32 | 100 MHz clock) at the end of block 1, 8, 32, 96
    lo = lambda x: (x & 0xffffffff).astype(np.int64)
    hi = lambda x: (x >> 32).astype(np.int64)
    nb = w[:, 4].astype(float)
    pts = [(1, 1), (5, 8), (6, 32), (7, 96)]
    for (sa, na), (sb, nb_) in zip(pts[:-1], pts[1:]):
        dt = ((lo(w[:, sb]) - lo(w[:, sa])) % (1 << 32)) / 100.0
        dc = (hi(w[:, sb]) - hi(w[:, sa])) % (1 << 32)
        print(f"blocks {na + 1:3d}-{nb_:3d}: us per block", q(dt / (nb_ - na)), "  shader clock MHz", q(dc / np.maximum(dt, 1e-9)))
    dt = ((lo(w[:, 3]) - lo(w[:, 7])) % (1 << 32)) / 100.0
    print("blocks  97-   : us per block", q(dt / np.maximum(nb - 96, 1)))
    print("first block ends (us after entry)", q(((lo(w[:, 1]) - lo(w[:, 0])) % (1 << 32)) / 100.0))
    sys.exit(0)
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
wg, wv = idx // WPB, idx % WPB
for name, key in (("workgroup % 8 (XCD)", wg % 8), ("wave of the workgroup", wv), ("workgroup // 256 (round of the launch)", wg // 256)):
    print(name + ":", "  ".join(f"{g}: {d[key == g].mean():.0f}" for g in np.unique(key)))
per_wg = np.array([d[wg == g].mean() for g in np.unique(wg)])
print("workgroup means: p1..p99", q(per_wg), " spread inside a workgroup (mean of max - min):", float(np.mean([np.ptp(d[wg == g]) for g in np.unique(wg)])))
