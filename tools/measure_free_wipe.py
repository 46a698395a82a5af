#!/usr/bin/env python3
"""What a large hipFree costs the next measurement: the scan of 100M f32 rows (HBM-bound, ~22.7 ms a pass) timed
pass by pass before and after device memory is given back.  On this driver freed VRAM is cleared in the background
at ~34 GB/s, and while that runs the scan loses ~2.6 % of its bandwidth (profiles/r02_free_wipe.txt).  bench.py and
tools/profile_round.sh wait the clear out (bench.settle) before they time anything after a large free.

    python tools/measure_free_wipe.py > gpurun_out/free_wipe.log
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import perceive_amd as pa  # noqa: E402


def main():
    ctx = pa.Context(0)
    rows = 100_000_000
    s = pa.Searcher(ctx, 384, "cosine")
    s.set_screening_copy("off")
    s.add_synthetic(1, rows, 0x5EED)
    s.finalize()
    s.set_kernel("mfma")
    rng = np.random.default_rng(1)

    def run(tag, secs):
        t0 = time.perf_counter()
        out = []
        while time.perf_counter() - t0 < secs:
            s.search_vectors(None, 10, rng.standard_normal((64, 384)).astype(np.float32))
            out.append((round(time.perf_counter() - t0, 2), round(s.last_stats()["scan_ms"], 2)))
        print(tag, "(seconds since the free, scan kernel ms):", out[::4], flush=True)

    run("nothing freed", 1.0)
    s.set_screening_copy("int8")
    s.finalize()
    s.set_screening_copy("off")  # drops the 38.8 GB int8 copy
    s.finalize()
    run("38.8 GB freed", 6.0)
    s2 = pa.Searcher(ctx, 384, "cosine")
    s2.set_screening_copy("off")
    s2.add_synthetic(1, 50_000_000, 3)
    s2.finalize()
    s2.close()
    run("76.8 GB freed", 10.0)
    s.close()
    ctx.close()


if __name__ == "__main__":
    main()
