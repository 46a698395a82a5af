#!/usr/bin/env python3
"""Text -> embedding throughput (Model::encode, model.rs:176-179): host tokenization (threaded C++ WordPiece) +
GPU encoder, 256 synthetic documents of ~230 words from the test vocabulary (no real corpus offline)."""
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
    ap.add_argument("--docs", type=int, default=256)
    ap.add_argument("--words", type=int, default=230)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--compute", default="f32", choices=["f32", "bf16x3", "f16x2"])
    a = ap.parse_args()
    vocab = os.path.join(ROOT, "tests", "golden", "tokenizer_vocab.txt")
    tok = pa.BertTokenizer(vocab)
    words = [w.strip() for w in open(vocab) if w.strip().isalpha() and len(w.strip()) > 3][:3000]
    rng = np.random.default_rng(0)
    docs = [" ".join(rng.choice(words, a.words)) for _ in range(a.docs)]
    ctx = pa.Context(0)
    d = pa.minilm_l6_desc(a.compute)
    d.vocab_size = tok.vocab_size
    m = pa.Model(ctx, d, synthetic_seed=1, tokenizer=tok)
    m.encode(docs)
    tk, en = [], []
    for _ in range(a.steps):
        t0 = time.perf_counter()
        ids, mask = m.tokenize(docs)
        t1 = time.perf_counter()
        m.encode_tokens(ids, mask)
        t2 = time.perf_counter()
        tk.append(t1 - t0)
        en.append(t2 - t1)
    tot = np.mean(tk) + np.mean(en)
    print(json.dumps({"metric": "documents/sec (text -> embedding)", "value": a.docs / tot, "unit": "docs/s",
                      "tokenize_ms": 1e3 * float(np.mean(tk)), "encode_ms": 1e3 * float(np.mean(en)), "tokens_per_doc": int(mask.sum() / a.docs),
                      "config": {"workload": f"{a.docs} documents x ~{a.words} words, MiniLM-L6 shape, {a.compute}"}}))
    m.close()
    ctx.close()


if __name__ == "__main__":
    main()
