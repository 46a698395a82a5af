"""CPU: the encoder oracle (oracle/encoder.c) against the committed Hugging Face BertModel vectors
(tests/golden/gen_encoder_golden.py)."""
import os

import numpy as np
import pytest


def load_tiny(golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_tiny.npz"))
    v, h, ly, nh, it, mp = [int(x) for x in g["desc"]]
    desc = dict(vocab=v, hidden=h, layers=ly, heads=nh, inter=it, max_pos=mp, eps=1e-12, pooling=0, normalize=1)
    weights = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    return g, desc, weights


def test_encoder_matches_hf_bert(oracle, golden_dir):
    g, desc, weights = load_tiny(golden_dir)
    out, hidden = oracle.encode_tokens(desc, weights, g["ids"], g["mask"], want_hidden=True)
    m = g["mask"].astype(bool)
    # hidden states of unmasked tokens, every layer (padded positions attend differently in HF
    # (dtype-min vs -10000) only when a whole row is masked, which never happens here)
    for ly in range(desc["layers"] + 1):
        np.testing.assert_allclose(hidden[ly][m], g["hidden"][ly][m], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out, g["normed"], rtol=0, atol=1e-5)
    d2 = dict(desc, normalize=0)
    out2, _ = oracle.encode_tokens(d2, weights, g["ids"], g["mask"])
    np.testing.assert_allclose(out2, g["mean"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-6)


def test_encoder_padding_invariance(oracle, golden_dir):
    # right-padding a batch further must not change the embeddings of the real tokens (mask works)
    g, desc, weights = load_tiny(golden_dir)
    ids, mask = g["ids"][:3], g["mask"][:3]
    a, _ = oracle.encode_tokens(desc, weights, ids, mask)
    ids2 = np.concatenate([ids, np.zeros((3, 8), np.int64)], 1)
    mask2 = np.concatenate([mask, np.zeros((3, 8), np.int64)], 1)
    b, _ = oracle.encode_tokens(desc, weights, ids2, mask2)
    np.testing.assert_allclose(a, b, atol=2e-6)


def test_pooling_modes_and_dense(oracle, golden_dir):
    g, desc, weights = load_tiny(golden_dir)
    _, hidden = oracle.encode_tokens(desc, weights, g["ids"], g["mask"], want_hidden=True)
    last = hidden[-1]
    cls, _ = oracle.encode_tokens(dict(desc, pooling=1, normalize=0), weights, g["ids"], g["mask"])
    np.testing.assert_allclose(cls, last[:, 0], atol=1e-6)
    mx, _ = oracle.encode_tokens(dict(desc, pooling=2, normalize=0), weights, g["ids"], g["mask"])
    ref = np.where(g["mask"][..., None].astype(bool), last, -1e9).max(1)
    np.testing.assert_allclose(mx, ref, atol=1e-6)
    rng = np.random.default_rng(0)
    w2 = dict(weights)
    w2["dense.linear.weight"] = rng.standard_normal((64, desc["hidden"])).astype(np.float32) * 0.1
    w2["dense.linear.bias"] = rng.standard_normal(64).astype(np.float32) * 0.1
    dn, _ = oracle.encode_tokens(dict(desc, normalize=0, dense_out=64, dense_act=1), w2, g["ids"], g["mask"])
    np.testing.assert_allclose(dn, np.tanh(g["mean"] @ w2["dense.linear.weight"].T + w2["dense.linear.bias"]), atol=2e-5)
