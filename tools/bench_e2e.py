#!/usr/bin/env python3
"""End to end (BASELINE configs[4], one GPU): encode a batch of token sequences with the
all-MiniLM-L6-v2-shaped encoder (seq_len 256, batch 256, seeded synthetic weights), then use the 256
embeddings as queries of an exact top-10 scan over the synthetic corpus (4 passes of 64 queries).
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import perceive_amd as pa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--compute", default="f32", choices=["f32", "bf16x3"])
    a = ap.parse_args()
    ctx = pa.Context(0)
    m = pa.Model(ctx, pa.minilm_l6_desc(a.compute), synthetic_seed=1)
    s = pa.Searcher(ctx, 384, "cosine")
    s.add_synthetic(1, a.rows, 0x5EED, normalize=True)  # MiniLM embeddings are unit-norm
    s.finalize()
    rng = np.random.default_rng(0)
    ids = rng.integers(1000, 30000, (a.batch, a.seq)).astype(np.int64)
    mask = np.ones_like(ids)
    enc_ms, scan_ms = [], []
    for i in range(a.warmup + a.steps):
        t0 = time.perf_counter()
        emb = m.encode_tokens(ids, mask)
        t1 = time.perf_counter()
        out = s.search_vectors(None, 10, emb)
        t2 = time.perf_counter()
        if i >= a.warmup:
            enc_ms.append(1e3 * (t1 - t0))
            scan_ms.append(1e3 * (t2 - t1))
    tot = np.mean(enc_ms) + np.mean(scan_ms)
    print(json.dumps({
        "metric": "end-to-end queries/sec (encode + exact top-10 scan)", "value": a.batch / (tot * 1e-3), "unit": "queries/s",
        "ms_per_step": tot, "encode_ms": float(np.mean(enc_ms)), "scan_ms": float(np.mean(scan_ms)),
        "config": {"workload": f"encode batch={a.batch} x seq_len={a.seq} (MiniLM-L6 shape, f32) + scan {a.rows} x 384, top-10, 1 MI355X"},
        "sample_ids": [int(x) for x in out[0][0][:3]],
    }))
    s.close()
    m.close()
    ctx.close()


if __name__ == "__main__":
    main()
