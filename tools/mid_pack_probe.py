#!/usr/bin/env python3
"""How long the mid copy takes to build with nothing else running (pcv_searcher_set_mid_copy ON + finalize):
    python tools/mid_pack_probe.py [rows] [dim]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import perceive_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
ctx = pa.Context(0)
s = pa.Searcher(ctx, d, "cosine")
s.add_synthetic(1, n, 0x5EED)
s.finalize()
ctx.synchronize()
for rep in range(3):
    s.set_mid_copy("on")
    t0 = time.perf_counter()
    s.finalize()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    dp = (d + 63) // 64 * 64
    print(f"rows {n} dim {d}: mid copy built in {1e3 * dt:.1f} ms (allocation included) = {n * dp * 6 / dt / 1e12:.2f} TB/s of reads + writes")
    s.set_mid_copy("off")
    s.finalize()
    ctx.synchronize()
    time.sleep(3)
s.close()
ctx.close()
