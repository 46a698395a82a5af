#!/usr/bin/env python3
"""Encoder throughput (BASELINE configs[4]: all-MiniLM-L6-v2 shape, seq_len=256, batch=256), seeded
synthetic weights.  Prints one JSON line: tokens/s and achieved TFLOP/s against the exact-f32 MFMA
peak (157.3 TF; the reference computes in f32)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import perceive_amd as pa  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ragged", action="store_true", help="lengths U[32, seq] instead of all-ones masks")
    ap.add_argument("--compute", default="f32", choices=["f32", "bf16x3", "f16x2"])
    ap.add_argument("--model", default="minilm-l6", choices=["minilm-l6", "bert-base"],
                    help="minilm-l6: 6 x 384, 12 heads, FFN 1536 (BASELINE configs[4]); bert-base: 12 x 768, 12 heads, FFN 3072 "
                         "(the shape of the reference's default MsMarcoBertBaseDotV5: CLS pooling, no normalisation)")
    a = ap.parse_args()
    ctx = pa.Context(0)
    if a.model == "bert-base":
        desc = pa.make_desc(30522, 768, 12, 12, 3072, 512, pooling="cls", normalize=False, max_seq_length=512, compute=a.compute)
    else:
        desc = pa.minilm_l6_desc(a.compute)
    m = pa.Model(ctx, desc, synthetic_seed=1)
    rng = np.random.default_rng(0)
    ids = rng.integers(1000, 30000, (a.batch, a.seq)).astype(np.int64)
    mask = np.ones((a.batch, a.seq), np.int64)
    if a.ragged:
        for b, n in enumerate(rng.integers(32, a.seq + 1, a.batch)):
            mask[b, n:] = 0
        ids *= mask
    for _ in range(a.warmup):
        m.encode_tokens(ids, mask)
    ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.encode_tokens(ids, mask)
        ms.append(m.last_stats()["total_ms"])
    wall = time.perf_counter() - t0
    st = m.last_stats()
    dev_ms = float(np.mean(ms))
    tf = st["flops"] / (dev_ms * 1e-3) / 1e12
    print(json.dumps({
        "metric": f"encoder tokens/sec ({a.model} shape)", "value": a.batch * a.seq * a.steps / wall,
        "unit": "tokens/s", "ms_per_step": 1e3 * wall / a.steps, "device_ms": dev_ms, "dtype": "f32",
        "config": {"workload": f"encode batch={a.batch} seq_len={a.seq}" + (" ragged" if a.ragged else ""), "compute": a.compute,
                   "shape": a.model},
        "roofline": {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / F32_MFMA_PEAK_TFLOPS, "flops_per_step": st["flops"]},
        "docs_per_s": a.batch * a.steps / wall,
    }))
    m.close()
    ctx.close()


if __name__ == "__main__":
    main()
