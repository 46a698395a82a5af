#!/usr/bin/env python3
"""A/B of scan kernel variants in ONE process on ONE corpus (guide rule 24: interleaved rounds, median and min):

    python tools/ab_scan.py --rows 100000000 --dim 384 --batch 256 --flags 0 0x20000000 [--metric dot --amplitude 0.5 2]

Each variant is a PCV_SCAN_FLAGS word (csrc/scan.h) set through pcv_searcher_set_tuning; prints the scan kernel's hipEvent
time per variant and checks that all variants return the same hits."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import perceive_amd as pa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, nargs="+", default=[64])
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--amplitude", type=float, nargs=2, default=None)
    ap.add_argument("--clustered", action="store_true")
    ap.add_argument("--flags", nargs="+", default=["0"])
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--screen", default="auto")
    ap.add_argument("--mid", default="auto", choices=["auto", "on", "off"], help="the row-major 16-bit mid copy (pcv_searcher_set_mid_copy)")
    args = ap.parse_args()
    ctx = pa.Context(0)
    s = pa.Searcher(ctx, args.dim, args.metric)
    if args.screen != "auto":
        s.set_screening_copy(args.screen)
    s.set_mid_copy(args.mid)
    ncl = max(1, args.rows // 20_000) if args.clustered else 0
    s.add_synthetic(1, args.rows, 0x5EED, n_clusters=ncl, noise=0.004 if args.clustered else 0.0, amplitude=args.amplitude)
    s.finalize()
    rng = np.random.default_rng(1)
    flags = [int(f, 0) for f in args.flags]
    for B in args.batch:
        if args.clustered:
            probe = s.get_rows(rng.integers(0, args.rows, (args.rounds + 2) * B))[0]
            qs = (probe + 0.002 * rng.standard_normal(probe.shape).astype(np.float32)).reshape(args.rounds + 2, B, args.dim)
        else:
            qs = rng.standard_normal((args.rounds + 2, B, args.dim)).astype(np.float32)
        times = {f: [] for f in flags}
        passes = {f: [] for f in flags}
        stats = {}
        for r in range(args.rounds + 2):
            ref = None
            for f in flags:
                s.set_tuning(f)
                ids, sc, _ = s.search_vectors(None, args.k, qs[r])
                if r < 2:
                    s.wait_background()  # (AUTO's mid copy is built beside the searches: the timed rounds are the steady state)
                st = s.last_stats()
                if ref is None:
                    ref = ids
                elif not (ids == ref).all():
                    print(f"MISMATCH batch {B} flags {f:#x} round {r}", flush=True)
                if r >= 2:
                    times[f].append(st["scan_ms"] / max(1, st["scan_launches"]))
                    passes[f].append(st["total_ms"] / max(1, st["scan_launches"]))
                    stats[f] = st
        for f in flags:
            t, st = np.array(times[f]), stats[f]
            gb = st["bytes_streamed"] / max(1, st["scan_launches"]) / 1e9
            print(f"rows={args.rows} dim={args.dim} {args.metric} B={B} flags={f:#x}: kernel median {np.median(t):.3f} ms  min {t.min():.3f}  "
                  f"pass {np.median(passes[f]):.3f} ms  {gb / np.median(t):.0f} GB/s = {gb / np.median(t) / 8:.3f} of 8 TB/s  launches {st['scan_launches']}  "
                  f"cand/q {st['candidates'] / B:.0f} coarse/q {st['coarse_survivors'] / B:.0f} copy {st['screening_copy']} mid {st['mid_copy']} mid/q {st['mid_survivors'] / B:.0f}", flush=True)
    s.close()
    ctx.close()


if __name__ == "__main__":
    main()
