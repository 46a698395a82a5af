#!/usr/bin/env python3
"""What the search calls around AUTO's decision to build the mid copy take (the build runs beside them, searcher.cpp: start_mid_build):
    python tools/mid_async_probe.py [rows, default 50000000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import perceive_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = pa.Context(0)
s = pa.Searcher(ctx, 384, "cosine")
s.add_synthetic(1, n, 0x5EED, n_clusters=n // 20_000, noise=0.004)
s.finalize()
rng = np.random.default_rng(5)
probe = s.get_rows(rng.integers(0, n, 64))[0]
q = (probe + 0.002 * rng.standard_normal(probe.shape)).astype(np.float32)
s.set_mid_copy("off")
t = []
for _ in range(6):
    t0 = time.perf_counter()
    s.search_vectors(None, 10, q)
    t.append(time.perf_counter() - t0)
print("rows", n, "call without the copy: %.3f ms" % (1e3 * np.median(t[2:])))
s.set_mid_copy("auto")
for i in range(60):
    t0 = time.perf_counter()
    s.search_vectors(None, 10, q)
    dt = time.perf_counter() - t0
    st = s.last_stats()
    print("call %d: %.3f ms  mid_copy %d  kernel %.3f ms" % (i, 1e3 * dt, st["mid_copy"], st["scan_ms"]))
    if st["mid_copy"] == 1 and i > 3:
        break
for _ in range(3):
    t0 = time.perf_counter()
    s.search_vectors(None, 10, q)
    print("with the copy: %.3f ms" % (1e3 * (time.perf_counter() - t0)))
s.close()
ctx.close()
