"""GPU, BASELINE.json's full size (100M x 384 f32 = 153.6 GB in HBM): the oracle cannot scan this in
seconds, so parity is shown through size-independent properties:
  * rows planted by value (the query itself, taken from known positions) come back at rank 1 with
    cosine 1, wherever they sit in the corpus (start, 2^32-work-item boundary, tail);
  * the two independent kernels (wave-reduction f32, MFMA bf16 screen) return identical ids;
  * every returned score is re-derived by the oracle from the row read back from HBM, the list is
    sorted, and a random sample of other rows never beats the k-th score;
  * splitting the corpus into shards and merging gives the same answer as the unsharded scan.
"""
import numpy as np
import pytest

import perceive_amd as pa

pytestmark = pytest.mark.gpu

N = 100_000_000
D = 384
SEED = 0xC0FFEE


@pytest.fixture(scope="module")
def big(ctx):
    s = pa.Searcher(ctx, D, "cosine")
    s.add_synthetic(1, N, SEED)
    s.finalize()
    assert s.num_rows == N
    yield s
    s.close()


def test_planted_rows_rank_first(big, oracle):
    pos = np.array([0, 31, 32, 10_485_760, 44_444_444, N - 33, N - 1], np.int64)
    q = np.stack([oracle.synth_rows(SEED, int(p), 1, D)[0] for p in pos])
    for kernel in ("wave", "mfma"):
        big.set_kernel(kernel)
        ids, sc, cnt = big.search_vectors(None, 10, q)
        np.testing.assert_array_equal(ids[:, 0], pos)
        np.testing.assert_allclose(sc[:, 0], 1.0, atol=1e-6)
        assert (sc[:, 1] < 0.5).all() and (cnt == 10).all()
    big.set_kernel("auto")


def test_kernels_agree_and_scores_verify(big, oracle):
    q = oracle.synth_rows(SEED + 1, 0, 8, D)
    big.set_kernel("mfma")
    ids_m, sc_m, _ = big.search_vectors(None, 10, q)
    st = big.last_stats()
    assert st["rows_scanned"] == N and st["overflow_reruns"] == 0 and st["candidates"] < 8 * 5000
    big.set_kernel("wave")
    ids_w, sc_w, _ = big.search_vectors(None, 10, q)
    big.set_kernel("auto")
    np.testing.assert_array_equal(ids_m, ids_w)
    np.testing.assert_array_equal(sc_m, sc_w)  # both come from the same f64 rescoring
    rows, rid = big.get_rows(ids_m.reshape(-1))
    np.testing.assert_array_equal(rid, ids_m.reshape(-1))
    rng = np.random.default_rng(0)
    sample = rng.integers(0, N, 4096)
    srows, _ = big.get_rows(sample)
    for b in range(q.shape[0]):
        ref = np.array([oracle.canonical_score(q[b], rows[b * 10 + j]) for j in range(10)])
        np.testing.assert_allclose(sc_m[b], ref.astype(np.float32), atol=1e-7)
        assert (np.diff(ref) <= 0).all()
        others = np.array([oracle.canonical_score(q[b], r) for r in srows[:512]])
        beat = others > ref[-1]
        assert set(sample[:512][beat]) <= set(ids_m[b])  # a sampled row above the k-th score must be in the list


def test_sharded_equals_whole(ctx, big, oracle):
    q = oracle.synth_rows(SEED + 2, 0, 4, D)
    ids, sc, _ = big.search_vectors(None, 10, q)
    # the same rows as two extra shard searchers would need another 153 GB; instead search two source
    # filters of a 3-source copy at 1/10 scale and compare with its own unsharded result
    n = 10_000_000
    whole = pa.Searcher(ctx, D, "cosine")
    whole.add_synthetic(1, n, SEED)
    whole.finalize()
    w_ids, w_sc, _ = whole.search_vectors(None, 10, q)
    whole.close()
    lists = ctx.alloc(3 * 4 * 10 * 24)
    shards = []
    for r in range(3):
        lo, hi = pa.shard_bounds(n, r, 3)
        s = pa.Searcher(ctx, D, "cosine")
        s.add_synthetic(1, hi - lo, SEED, first_row=lo)
        s.finalize()
        s.set_shard_offset(lo)
        s.search_device(None, 10, q, lists + r * 4 * 10 * 24)
        shards.append(s)
    m_ids, m_sc, _ = pa.merge_topk(ctx, "cosine", D, lists, 3, 4, 10)
    np.testing.assert_array_equal(m_ids, w_ids)
    np.testing.assert_array_equal(m_sc, w_sc)
    ctx.free(lists)
    for s in shards:
        s.close()
    # and the 100M result restricted to the first 10M rows is consistent with the 10M corpus
    for b in range(4):
        small = [i for i in ids[b] if i < n]
        assert small == [i for i in w_ids[b] if i in set(small)]
