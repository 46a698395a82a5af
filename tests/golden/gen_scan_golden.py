"""Generates tests/golden/scan_*.npz with PyTorch-CPU executing the op sequence of the reference's
brute-force similarity functions (crates/perceive-core/lib.rs:63-77):

    dot_product(a, b)                 = a.matmul(b.transpose(0, 1))
    cosine_similarity_single_query    = (q / q.linalg_norm(2, [0], True)) . (M / M.linalg_norm(2, [1], True))^T
    cosine_similarity_multi_query     = same with [B, D] queries normalised over dim 1

The reference itself is Rust (tch-rs -> libtorch) and cannot be built here; torch's ATen CPU
kernels are the same operator family tch binds, so these vectors pin the CPU oracle to the
reference's *arithmetic* (f32, same ops), not to a run of the reference.  Run once, offline:

    python tests/golden/gen_scan_golden.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def dot_product(a, b):
    return a.matmul(b.transpose(0, 1))


def cosine_similarity_single_query(query, matches):
    query = query / torch.linalg.norm(query, 2.0, [0], True, dtype=torch.float32)
    matches = matches / torch.linalg.norm(matches, 2.0, [1], True, dtype=torch.float32)
    return dot_product(query, matches)


def cosine_similarity_multi_query(set1, set2):
    set1 = set1 / torch.linalg.norm(set1, 2.0, [1], True, dtype=torch.float32)
    set2 = set2 / torch.linalg.norm(set2, 2.0, [1], True, dtype=torch.float32)
    return dot_product(set1, set2)


def canonical_f64(q, m):
    """f64 evaluation of the same quantity (ranking reference; order of summation is torch's)."""
    q64, m64 = q.double(), m.double()
    return (q64 @ m64.T) / (q64.norm(dim=1, keepdim=True) * m64.norm(dim=1)[None, :])


def topk_desc_stable(scores, k):
    # descending score, ties -> lower index
    order = np.lexsort((np.arange(scores.shape[-1])[None, :].repeat(scores.shape[0], 0), -scores), axis=-1)
    return order[:, :k]


def main():
    torch.manual_seed(0x5EED)
    torch.set_num_threads(1)
    N, D, B, K = 1000, 384, 64, 10
    corpus = torch.randn(N, D, dtype=torch.float32)
    # edge rows: an exact duplicate pair, a scaled copy (same cosine, different norm), a zero row
    corpus[777] = corpus[123]
    corpus[778] = corpus[123] * 3.0
    corpus[500] = 0.0
    queries = torch.randn(B, D, dtype=torch.float32)
    queries[5] = corpus[123] + 0.05 * torch.randn(D)  # near-duplicate query -> tie among 123/777/778
    queries[6] = corpus[42] * 0.5                     # cosine exactly ~1 with row 42

    multi = cosine_similarity_multi_query(queries, corpus).numpy()        # [B, N] (NaN in column 500)
    single = cosine_similarity_single_query(queries[0], corpus).numpy()   # [N]
    dots = dot_product(queries, corpus).numpy()
    c64 = canonical_f64(queries, corpus).numpy()

    valid = np.ones(N, dtype=bool)
    valid[500] = False
    c64m = np.where(valid[None, :], c64, -np.inf)
    f32m = np.where(valid[None, :], np.nan_to_num(multi, nan=-np.inf), -np.inf)
    top_f64 = topk_desc_stable(c64m, K + 1)
    top_f32 = topk_desc_stable(f32m, K)
    # gap between rank K and K+1 in f64: where it is < 1e-6 an f32 evaluation may legitimately
    # order the boundary differently
    gap = np.take_along_axis(c64m, top_f64[:, K - 1 : K], 1)[:, 0] - np.take_along_axis(c64m, top_f64[:, K : K + 1], 1)[:, 0]

    np.savez_compressed(
        os.path.join(HERE, "scan_n1000_d384.npz"),
        corpus=corpus.numpy(), queries=queries.numpy(),
        cos_multi_f32=multi, cos_single_q0_f32=single, dot_f32=dots,
        cos_f64=c64, topk_f64=top_f64[:, :K].astype(np.int64), topk_f32=top_f32.astype(np.int64),
        rank_gap_f64=gap, k=np.int64(K),
    )

    # small odd-shaped case: D not a multiple of 64, N not a multiple of 32, few queries
    torch.manual_seed(7)
    c2 = torch.randn(77, 100, dtype=torch.float32)
    q2 = torch.randn(3, 100, dtype=torch.float32)
    m2 = cosine_similarity_multi_query(q2, c2).numpy()
    c642 = canonical_f64(q2, c2).numpy()
    np.savez_compressed(
        os.path.join(HERE, "scan_n77_d100.npz"),
        corpus=c2.numpy(), queries=q2.numpy(), cos_multi_f32=m2, dot_f32=dot_product(q2, c2).numpy(),
        cos_f64=c642, topk_f64=topk_desc_stable(c642, 5).astype(np.int64), k=np.int64(5),
    )
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
