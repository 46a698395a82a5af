"""GPU parity tests of the similarity scan, through the C ABI (perceive_amd -> libperceive_hip.so),
against the CPU oracle and the committed PyTorch-CPU vectors.

Bars (BASELINE.json north_star): top-k doc indices bit-exact, cosine scores within 1e-4 (f32).
"""
import os

import numpy as np
import pytest

import perceive_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: "cosine scores within 1e-4 f32"


def build(ctx, corpus, metric="cosine", ids=None, source=1, kernel="auto", screen=None):
    s = pa.Searcher(ctx, corpus.shape[1], metric)
    if screen is not None:
        s.set_screening_copy(screen)
    s.add_rows(source, corpus, ids)
    s.finalize()
    s.set_kernel(kernel)
    return s


@pytest.fixture(scope="module")
def g1000(golden_dir):
    return np.load(os.path.join(golden_dir, "scan_n1000_d384.npz"))


@pytest.fixture(scope="module")
def g77(golden_dir):
    return np.load(os.path.join(golden_dir, "scan_n77_d100.npz"))


@pytest.mark.parametrize("screen", ["int8", "bf16", "off"])
@pytest.mark.parametrize("kernel,B", [("auto", 1), ("auto", 3), ("auto", 64), ("wave", 64), ("mfma", 1), ("mfma", 33)])
def test_golden_1000(ctx, oracle, g1000, kernel, B, screen):
    # screen: which resident screening copy of the rows the coarse screen streams (none: the f32 rows themselves)
    k = int(g1000["k"])
    s = build(ctx, g1000["corpus"], kernel=kernel, screen=screen)
    q = g1000["queries"][:B]
    ids, scores, counts = s.search_vectors(None, k, q)
    assert (counts == k).all()
    # indices: bit-exact against the torch-f64 ranking stored in the fixture and the C oracle
    np.testing.assert_array_equal(ids, g1000["topk_f64"][:B])
    opos, osc, _ = oracle.topk(q, g1000["corpus"], k)
    np.testing.assert_array_equal(ids, opos)
    # scores: within 1e-4 of the reference-shaped f32 torch result (and ~1e-7 of the f64 value)
    ref32 = np.take_along_axis(g1000["cos_multi_f32"][:B], ids, 1)
    assert np.abs(scores - ref32).max() < TOL
    np.testing.assert_allclose(scores, osc.astype(np.float32), rtol=0, atol=1e-7)
    st = s.last_stats()
    mfma = kernel == "mfma" or (kernel == "auto" and (B > 4 or screen != "off"))  # with copies the MFMA kernel streams fewer bytes
    assert st["kernel_used"] == (2 if mfma else 1)
    assert st["screening_copy"] == ({"int8": 2, "bf16": 1, "off": 0}[screen] if mfma else 0)
    nblk = st["scan_launches"] * ((1000 + 31) // 32)  # per 32-row block: f32 pieces + 32 scales | bf16 pieces | int8 pieces + the block's scale
    assert st["bytes_streamed"] == nblk * {0: 384 * 4 * 32 + 128, 1: 384 * 2 * 32, 2: 384 * 32 + 4}[st["screening_copy"]]
    assert st["rows_scanned"] >= 1000 and st["overflow_reruns"] == 0
    s.close()


def test_golden_odd_shape(ctx, oracle, g77):
    # D=100 (padded to 128 in HBM), N=77 (3 blocks, last one ragged), 3 queries, both kernels
    k = int(g77["k"])
    for kernel in ("wave", "mfma"):
        s = build(ctx, g77["corpus"], kernel=kernel)
        ids, scores, counts = s.search_vectors(None, k, g77["queries"])
        np.testing.assert_array_equal(ids, g77["topk_f64"])
        ref = np.take_along_axis(g77["cos_multi_f32"], ids, 1)
        assert np.abs(scores - ref).max() < TOL
        s.close()


def test_ties_zero_rows_and_k_larger_than_valid(ctx, oracle, g1000):
    corpus = g1000["corpus"]
    s = build(ctx, corpus)
    # query 5: rows 123 / 777 (exact duplicate) / 778 (3x copy) lead; duplicate resolves to lower position
    ids, scores, counts = s.search_vectors(None, 3, g1000["queries"][5:6])
    opos, _, _ = oracle.topk(g1000["queries"][5:6], corpus, 3)
    np.testing.assert_array_equal(ids, opos)
    assert list(ids[0]).index(123) < list(ids[0]).index(777)
    # zero row 500 is never returned, even when k covers the whole corpus (k capped at 128 per call)
    ids, scores, counts = s.search_vectors(None, 128, g1000["queries"][:2])
    assert (counts == 128).all() and 500 not in ids
    opos, _, _ = oracle.topk(g1000["queries"][:2], corpus, 128)
    np.testing.assert_array_equal(ids, opos)
    # more results than one pass ranks (128): the library goes over the rows again below the last hit of the pass before
    # (the reference has no limit on num_results, search.rs:157-182).  k = 1000 covers the whole 1000-row corpus: the zero
    # row has no score and is not returned, 999 hits; the order is the oracle's, duplicates (equal scores) by position.
    for kernel in ("wave", "mfma"):
        s.set_kernel(kernel)
        for k in (129, 300, 1000):
            ids, scores, counts = s.search_vectors(None, k, g1000["queries"][:3])
            opos, osc, ocnt = oracle.topk(g1000["queries"][:3], corpus, k)
            np.testing.assert_array_equal(counts, ocnt)
            np.testing.assert_array_equal(ids, opos)
            live = opos >= 0
            assert np.abs(scores[live] - osc.astype(np.float32)[live]).max() < 1e-6
            assert counts[0] == min(k, 999) and 500 not in ids
    s.close()
    # fewer valid rows than k
    m = np.zeros((5, 8), np.float32)
    m[1, 0] = 1.0
    m[3, 1] = 2.0
    s = build(ctx, m)
    ids, scores, counts = s.search_vectors(None, 4, np.array([[1, 1, 0, 0, 0, 0, 0, 0]], np.float32))
    assert counts[0] == 2 and list(ids[0]) == [1, 3, -1, -1]
    np.testing.assert_allclose(scores[0][:2], [2**-0.5, 2**-0.5], atol=1e-7)
    assert np.isnan(scores[0][2:]).all()
    # zero query: cosine undefined for every row -> nothing returned (reference: NaN then panic)
    ids, scores, counts = s.search_vectors(None, 2, np.zeros((1, 8), np.float32))
    assert counts[0] == 0
    s.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("screen", ["int8", "bf16", "off"])
def test_more_results_than_one_pass_ranks(ctx, oracle, metric, screen):
    # num_results > PCV_MAX_RESULTS (128): passes below the previous pass's last hit (scan.h, CeilRec), every screen form, both
    # metrics, two sources with rows of unequal norm, duplicated rows that straddle a pass boundary (equal scores: by position),
    # and a query whose rows run out before num_results.
    rng = np.random.default_rng(11)
    N, D, k = 40_000, 96, 300
    m = (rng.standard_normal((N, D)) * rng.uniform(0.5, 2.0, (N, 1))).astype(np.float32)
    q = rng.standard_normal((5, D)).astype(np.float32)
    # rows 1000..1009 are copies of the row that ranks 126th for query 0 under this metric: ranks 126..136 are one score
    s0 = build(ctx, m, metric=metric, screen=screen)
    first, _, _ = s0.search_vectors(None, 128, q[:1])
    s0.close()
    m[1000:1010] = m[first[0, 125]]
    s = pa.Searcher(ctx, D, metric)
    s.set_screening_copy(screen)
    s.add_rows(1, m[:25_000], np.arange(25_000, dtype=np.int64))
    s.add_rows(2, m[25_000:], np.arange(25_000, N, dtype=np.int64))
    s.finalize()
    om = {"cosine": 0, "dot": 1}[metric]
    for kernel in ("auto", "wave"):
        s.set_kernel(kernel)
        ids, scores, counts = s.search_vectors(None, k, q)
        opos, osc, ocnt = oracle.topk(q, m, k, metric=om)
        np.testing.assert_array_equal(ids, opos)
        assert (counts == k).all()
        assert s.last_stats()["scan_launches"] >= 3  # 128 + 128 + 44
    # one source only, and more results than it has rows
    sub = m[25_000:25_000 + 200]
    s2 = build(ctx, sub, metric=metric, screen=screen)
    ids, scores, counts = s2.search_vectors(None, 260, q[:2])
    opos, _, ocnt = oracle.topk(q[:2], sub, 260, metric=om)
    np.testing.assert_array_equal(ids, opos)
    np.testing.assert_array_equal(counts, ocnt)
    assert (counts == 200).all()
    s2.close()
    s.close()


def test_nan_and_inf_rows_are_skipped(ctx, oracle):
    rng = np.random.default_rng(3)
    m = rng.standard_normal((300, 64)).astype(np.float32)
    m[10, 3] = np.nan
    m[20, 5] = np.inf
    m[30] = 1e30  # huge but finite: |x|^2 overflows f32, fine in f64
    q = rng.standard_normal((6, 64)).astype(np.float32)
    for kernel in ("wave", "mfma"):
        s = build(ctx, m, kernel=kernel)
        ids, scores, counts = s.search_vectors(None, 7, q)
        opos, osc, ocnt = oracle.topk(q, m, 7)
        np.testing.assert_array_equal(ids, opos)
        np.testing.assert_allclose(scores, osc.astype(np.float32), atol=1e-7)
        assert 10 not in ids and 20 not in ids
        s.close()


def test_dot_metric_matches_search_vector(ctx, oracle):
    # the reference Searcher's convention: distance max(0, 1 - dot/len), ascending (search.rs:157-182,266-279)
    rng = np.random.default_rng(11)
    m = rng.standard_normal((5000, 384)).astype(np.float32) * rng.uniform(0.2, 3.0, (5000, 1)).astype(np.float32)
    ids = rng.permutation(100000)[:5000].astype(np.int64)
    src = np.where(np.arange(5000) < 3000, 7, 9)
    q = rng.standard_normal((5, 384)).astype(np.float32)
    for kernel in ("wave", "mfma"):
        s = pa.Searcher(ctx, 384, "dot")
        s.add_rows(7, m[src == 7], ids[src == 7])
        s.add_rows(9, m[src == 9], ids[src == 9])
        s.finalize()
        s.set_kernel(kernel)
        assert s.source_ids == [7, 9] and s.num_rows == 5000
        for sources in ([7], [9], [7, 9], None):
            got_ids, got_d, cnt = s.search_vectors(sources, 20, q)
            for b in range(q.shape[0]):
                oi, od = oracle.search_vector(q[b], m, ids, src, sources if sources else [], 20)
                np.testing.assert_array_equal(got_ids[b], oi)
                np.testing.assert_allclose(got_d[b], od, atol=1e-6)
                assert (np.diff(got_d[b]) >= 0).all()
        assert s.search_vectors([], 5, q)[2].sum() == 0  # empty filter matches nothing (search.rs:166)
        assert s.search_vectors([12345], 5, q)[2].sum() == 0
        # the single-vector API returns SearchItems like search.rs:171-174
        items = s.search_vector([9], 3, q[0])
        oi, od = oracle.search_vector(q[0], m, ids, src, [9], 3)
        assert [it.id for it in items] == list(oi)
        s.close()
    # clamp at distance 0 for dot >= len (search.rs:277): order still follows the dot product
    m2 = np.eye(8, dtype=np.float32) * np.arange(1, 9, dtype=np.float32)[:, None] * 10
    s = build(ctx, m2, metric="dot")
    q2 = np.ones((1, 8), np.float32)
    got_ids, got_d, _ = s.search_vectors(None, 8, q2)
    assert list(got_ids[0]) == [7, 6, 5, 4, 3, 2, 1, 0] and (got_d[0] == 0).all()
    s.close()


def test_build_from_row_stream_and_rebuild_source(ctx, oracle):
    rng = np.random.default_rng(5)
    m = rng.standard_normal((600, 384)).astype(np.float32)
    item_ids = np.arange(600) * 3 + 1
    src = np.where(np.arange(600) % 3 == 0, 1, 2)
    # rows as Searcher::build reads them: (items.id, source_id, embedding BLOB) search.rs:87-113
    rows = [(int(item_ids[i]), int(src[i]), pa.serialize_embedding(m[i])) for i in range(600)]
    s = pa.Searcher.build(ctx, rows, 384, metric="dot")
    q = rng.standard_normal(384).astype(np.float32)
    order = np.concatenate([np.nonzero(src == 1)[0], np.nonzero(src == 2)[0]])  # sources kept apart
    items = s.search_vector([1, 2], 10, q)
    oi, od = oracle.search_vector(q, m[order], item_ids[order], src[order], [1, 2], 10)
    assert [it.id for it in items] == list(oi)
    np.testing.assert_allclose([it.score for it in items], od, atol=1e-6)
    # rebuild_source (search.rs:58-79): source 2 replaced by new rows, source 1 untouched
    m_new = rng.standard_normal((50, 384)).astype(np.float32)
    new_rows = [(9000 + i, 2, m_new[i]) for i in range(50)] + [(777777, 1, m_new[0])]  # other sources ignored
    s.rebuild_source(new_rows, 2)
    assert s.num_rows == 200 + 50 and s.source_ids == [1, 2]
    allm = np.concatenate([m[src == 1], m_new])
    allids = np.concatenate([item_ids[src == 1], 9000 + np.arange(50)])
    allsrc = np.concatenate([np.full(200, 1), np.full(50, 2)])
    for sources in ([2], [1], [1, 2]):
        items = s.search_vector(sources, 10, q)
        oi, od = oracle.search_vector(q, allm, allids, allsrc, sources, 10)
        assert [it.id for it in items] == list(oi)
    # rebuilding with no rows leaves the source absent (search.rs:67-69)
    s.rebuild_source([], 2)
    assert s.source_ids == [1] and s.num_rows == 200
    s.hidden.add(1)  # pub field kept (search.rs:34); like the reference's search_vector it does not filter
    assert s.search_vector([1], 1, m[0])[0].id == 1  # row 0 (item id 1, source 1) still found
    s.close()


def test_synthetic_rows_bit_identical_to_oracle(ctx, oracle):
    for normalize in (False, True):
        s = pa.Searcher(ctx, 384, "cosine")
        s.add_synthetic(1, 5000, 0x5EED, first_row=1000, normalize=normalize)
        s.finalize()
        pos = np.array([0, 1, 31, 32, 33, 1023, 1024, 4999], np.int64)
        rows, ids = s.get_rows(pos)
        ref = oracle.synth_rows(0x5EED, 1000, 5000, 384, normalize)
        np.testing.assert_array_equal(rows.view(np.uint32), ref[pos].view(np.uint32))
        np.testing.assert_array_equal(ids, 1000 + pos)
        # and a search over them agrees with the oracle on the oracle-generated copy
        q = oracle.synth_rows(0x5EED + 1, 0, 5, 384)
        got, sc, _ = s.search_vectors(None, 10, q)
        opos, osc, _ = oracle.topk(q, ref, 10)
        np.testing.assert_array_equal(got, opos + 1000)
        np.testing.assert_allclose(sc, osc.astype(np.float32), atol=1e-7)
        s.close()


def test_synthetic_fill_beyond_2_pow_32_work_items(ctx, oracle):
    # 12M x 384 rows = 1.15e9 float4 pieces per 32-row block column... the fill kernel covers
    # 12M*96 = 1.15e9 (x32 lanes -> 3.7e10 work items): a launch sized in work items would wrap at 2^32
    N = 12_000_000
    s = pa.Searcher(ctx, 384, "cosine")
    s.add_synthetic(1, N, 99)
    s.finalize()
    pos = np.array([0, 10_485_759, 10_485_760, 10_500_000, N - 33, N - 1], np.int64)
    rows, ids = s.get_rows(pos)
    for i, p_ in enumerate(pos):
        np.testing.assert_array_equal(rows[i].view(np.uint32), oracle.synth_rows(99, int(p_), 1, 384)[0].view(np.uint32))
    # a row planted by value near the end must be found: query = that row -> cosine 1 at its position
    q = oracle.synth_rows(99, N - 5, 1, 384)
    got, sc, _ = s.search_vectors(None, 1, q)
    assert got[0, 0] == N - 5 and abs(sc[0, 0] - 1.0) < 1e-6
    got64, sc64, _ = s.search_vectors(None, 1, np.repeat(q, 5, 0))
    assert (got64[:, 0] == N - 5).all()
    s.close()


@pytest.mark.parametrize("B,kernel", [(1, "wave"), (4, "wave"), (8, "mfma"), (64, "mfma"), (100, "mfma"), (128, "mfma")])
def test_medium_random_vs_oracle(ctx, oracle, B, kernel):
    N = 200_000 if B <= 8 else 60_000
    s = pa.Searcher(ctx, 384, "cosine")
    s.add_synthetic(1, N, 42)
    s.finalize()
    s.set_kernel(kernel)
    q = oracle.synth_rows(43, 0, B, 384)
    ids, scores, counts = s.search_vectors(None, 10, q)
    ref = oracle.synth_rows(42, 0, N, 384)
    opos, osc, _ = oracle.topk(q, ref, 10)
    np.testing.assert_array_equal(ids, opos)
    np.testing.assert_allclose(scores, osc.astype(np.float32), atol=1e-7)
    st = s.last_stats()
    # the screen must be selective: survivors are a tiny fraction of the B*N pairs
    assert st["candidates"] < 0.02 * B * N, st
    s.close()


def test_candidate_overflow_reruns_and_stays_exact(ctx, oracle):
    # adversarial order: similarity grows with the row index, so every row beats the running k-th
    # best and the candidate lists overflow their initial capacity; the pass must be repeated with
    # larger lists and still return the exact answer.
    rng = np.random.default_rng(9)
    N, D = 40_000, 64
    q = rng.standard_normal(D).astype(np.float32)
    noise = rng.standard_normal((N, D)).astype(np.float32)
    t = np.linspace(-1, 1, N, dtype=np.float32)[:, None]
    m = (t * q[None, :] + 0.01 * noise).astype(np.float32)
    for kernel in ("wave", "mfma"):
        s = build(ctx, m, kernel=kernel)
        s.set_candidate_capacity(64)  # a massively parallel scan raises the threshold fast: start with short lists
        ids, scores, counts = s.search_vectors(None, 10, q[None, :])
        opos, osc, _ = oracle.topk(q[None, :], m, 10)
        np.testing.assert_array_equal(ids, opos)
        st = s.last_stats()
        assert st["overflow_reruns"] >= 1 and st["scan_launches"] == 1 + st["overflow_reruns"], st
        # the grown lists stay: the same search now fits
        ids2, _, _ = s.search_vectors(None, 10, q[None, :])
        np.testing.assert_array_equal(ids2, opos)
        assert s.last_stats()["overflow_reruns"] == 0
        s.close()


@pytest.mark.parametrize("B", [5, 64])
def test_survivor_ring_when_every_row_passes_the_coarse_screen(ctx, oracle, B):
    # The DRAIN form of the int8 scan (5..64 queries): streaming waves write coarse survivors into an LDS ring that one drain wave
    # per CU works off.  Here nearly every (row, query) pair survives the coarse test — the rows are one vector plus noise far
    # below the int8 margin, the queries are that vector — so every wave fills the ring at every block and has to wait for room,
    # the candidate lists overflow and the pass is repeated with longer ones: no hang, exact hits (ties by position).
    rng = np.random.default_rng(4)
    N, D = 48_000, 128
    v = rng.standard_normal(D).astype(np.float32)
    m = (v[None, :] + 1e-4 * rng.standard_normal((N, D))).astype(np.float32)
    m[1000:1010] = v  # ten exact copies: cosine 1, ranked by position
    q = np.repeat(v[None, :], B, 0) + (1e-5 * rng.standard_normal((B, D))).astype(np.float32)
    q[0] = v
    s = build(ctx, m, kernel="mfma", screen="int8")
    ids, scores, counts = s.search_vectors(None, 10, q)
    st = s.last_stats()
    opos, osc, _ = oracle.topk(q, m, 10)
    np.testing.assert_array_equal(ids, opos)
    assert list(ids[0]) == list(range(1000, 1010))
    assert np.abs(scores - osc.astype(np.float32)).max() < 1e-6
    assert st["coarse_survivors"] > 0.5 * N * B, st  # nearly every pair, at every launch
    assert st["overflow_reruns"] >= 1, st
    s.close()


def test_many_sources_one_launch(ctx, oracle):
    # 11 sources -> 11 segments: one launch walks them all through the device segment table
    rng = np.random.default_rng(21)
    s = pa.Searcher(ctx, 128, "cosine")
    parts = []
    for src in range(11):
        m = rng.standard_normal((100 + 7 * src, 128)).astype(np.float32)
        parts.append(m)
        s.add_rows(src, m, np.arange(m.shape[0]) + 1000 * src)
    s.finalize()
    allm = np.concatenate(parts)
    allids = np.concatenate([np.arange(p.shape[0]) + 1000 * i for i, p in enumerate(parts)])
    q = rng.standard_normal((3, 128)).astype(np.float32)
    ids, scores, _ = s.search_vectors(None, 12, q)
    opos, osc, _ = oracle.topk(q, allm, 12)
    np.testing.assert_array_equal(ids, allids[opos])
    assert s.last_stats()["scan_launches"] == 1
    s.close()


@pytest.mark.parametrize("kernel,B", [("wave", 2), ("mfma", 40)])
def test_forty_sources_three_incremental_adds_one_launch(ctx, oracle, kernel, B):
    # the reference keeps one index per source (search.rs:24-27,134-155) and a real database grows by
    # repeated scans: 40 sources x 3 add+finalize rounds.  Adds append in place while a segment has room,
    # one launch walks every segment, and the begin/end protocol works on the same searcher.
    rng = np.random.default_rng(1234)
    D, k = 128, 10
    s = pa.Searcher(ctx, D, "cosine")
    s.set_kernel(kernel)
    per_source = {src: [] for src in range(40)}
    next_id = 0
    for rnd in range(3):
        for src in range(40):
            n = int(rng.integers(5, 90)) * (rnd + 1)
            m = rng.standard_normal((n, D)).astype(np.float32)
            s.add_rows(src, m, np.arange(next_id, next_id + n))
            per_source[src].append((m, np.arange(next_id, next_id + n)))
            next_id += n
        s.finalize()
    allm = np.concatenate([m for src in range(40) for m, _ in per_source[src]])
    allids = np.concatenate([i for src in range(40) for _, i in per_source[src]])
    assert s.num_rows == allm.shape[0] and len(s.source_ids) == 40
    assert s.num_segments == 40  # not 120: the later adds went into the spare room of each source's first segment
    q = rng.standard_normal((B, D)).astype(np.float32)
    ids, sc, _ = s.search_vectors(None, k, q)
    opos, osc, _ = oracle.topk(q, allm, k)
    np.testing.assert_array_equal(ids, allids[opos])
    np.testing.assert_allclose(sc, osc.astype(np.float32), atol=1e-7)
    st = s.last_stats()
    assert st["scan_launches"] == (B + 3) // 4 if kernel == "wave" else st["scan_launches"] == 1
    # a source filter picks that source's segments only
    sub = [3, 17, 39]
    ids, _, _ = s.search_vectors(sub, k, q)
    subm = np.concatenate([m for src in sub for m, _ in per_source[src]])
    subids = np.concatenate([i for src in sub for _, i in per_source[src]])
    np.testing.assert_array_equal(ids, subids[oracle.topk(q, subm, k)[0]])
    # begin/end on all segments: queued without a host round trip, one launch
    if B <= 4 or kernel == "mfma":
        rec = B * k + 1
        d = ctx.alloc(rec * 24)
        s.search_device_begin(None, k, q, d)
        assert s.search_device_end() is False
        assert s.last_stats()["scan_launches"] == 1
        got, _, _, over = pa.merge_topk(ctx, "cosine", D, d, 1, B, k, flagged=True)
        assert over is False
        np.testing.assert_array_equal(got, allids[opos])
        ctx.free(d)
    s.close()


def test_reserve_gives_one_segment_and_tail_append(ctx, oracle):
    rng = np.random.default_rng(8)
    D = 64
    m = rng.standard_normal((5000, D)).astype(np.float32)
    s = pa.Searcher(ctx, D, "dot")
    s.reserve(7, 5000)
    for lo in range(0, 5000, 777):  # ragged chunks, block boundaries inside and across chunks
        s.add_rows(7, m[lo:lo + 777], 10_000 + np.arange(lo, min(lo + 777, 5000)))
    s.finalize()
    assert s.num_segments == 1 and s.num_rows == 5000
    rows, ids = s.get_rows(np.array([0, 31, 32, 776, 777, 778, 4999]))
    np.testing.assert_array_equal(rows, m[[0, 31, 32, 776, 777, 778, 4999]])
    np.testing.assert_array_equal(ids, 10_000 + np.array([0, 31, 32, 776, 777, 778, 4999]))
    q = rng.standard_normal((3, D)).astype(np.float32)
    got, _, _ = s.search_vectors([7], 9, q)
    np.testing.assert_array_equal(got, 10_000 + oracle.topk(q, m, 9, metric=1)[0])
    # implicit ids keep counting across adds
    s2 = pa.Searcher(ctx, D, "cosine")
    s2.add_rows(1, m[:100])
    s2.add_rows(1, m[100:250])
    s2.finalize()
    got, _, _ = s2.search_vectors(None, 5, q)
    np.testing.assert_array_equal(got, oracle.topk(q, m[:250], 5)[0])
    s.close()
    s2.close()


def test_streaming_ingest_keeps_host_memory_bounded(ctx):
    # Searcher::build streams rows out of SQLite (search.rs:87-113); the library must not keep a host copy:
    # 20M x 384 rows (30.7 GB) pushed through one 131072-row buffer, peak RSS of the process < 2 GB
    import subprocess
    import sys

    code = (
        "import sys, re, numpy as np; sys.path.insert(0, %r); import perceive_amd as pa\n"
        "ctx = pa.Context(0); s = pa.Searcher(ctx, 384, 'cosine')\n"
        "rng = np.random.default_rng(1); chunk = rng.standard_normal((131072, 384)).astype(np.float32)\n"
        "blob = chunk.tobytes(); N = 20_000_000; done = 0\n"
        "while done < N:\n"
        "    n = min(131072, N - done)\n"
        "    s.add_blobs(1 + done // 5_000_000, blob[: n * 1536], n, ids=np.arange(done, done + n))\n"
        "    done += n\n"
        "s.finalize()\n"
        "q = chunk[77:78] + 0.0\n"
        "ids, sc, cnt = s.search_vectors(None, 3, q)\n"
        # peak RSS of THIS process image: VmHWM (ru_maxrss also carries the parent's peak across fork + exec)
        "hwm = int(re.search(r'VmHWM:\\s+(\\d+) kB', open('/proc/self/status').read()).group(1))\n"
        "print('RESULT', s.num_rows, s.num_segments, int(ids[0, 0]) %% 131072, float(sc[0, 0]), hwm)\n"
    ) % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    rows, nseg, hit, score, maxrss_kb = int(line[1]), int(line[2]), int(line[3]), float(line[4]), int(line[5])
    assert rows == 20_000_000 and hit == 77 and abs(score - 1.0) < 1e-6
    assert nseg <= 4 * 8, nseg  # 4 sources of 5M rows, each a short chain of growing segments
    assert maxrss_kb < 2 * 1024 * 1024, f"peak RSS {maxrss_kb / 1e6:.2f} GB"


def test_clustered_rows_stay_exact_and_selective(ctx, oracle):
    # clustered corpus (centroid + small noise): thousands of rows sit inside the bf16 margin of the k-th
    # best.  The exact-f32 second screen keeps the lists short, nothing overflows, results are exact.
    N, D, B, k = 300_000, 384, 64, 10
    ncl, noise = 30, 0.004  # ~10^4 rows per cluster, cosines inside a cluster spread over ~0.006
    ref = oracle.synth_rows_clustered(0xABCD, 0, N, D, ncl, noise)
    s = pa.Searcher(ctx, D, "cosine")
    s.add_synthetic(1, N, 0xABCD, n_clusters=ncl, noise=noise)
    s.finalize()
    rows, _ = s.get_rows(np.array([0, 12345, N - 1]))
    np.testing.assert_array_equal(rows.view(np.uint32), ref[[0, 12345, N - 1]].view(np.uint32))
    q = oracle.synth_rows_clustered(0xABCD, N + 5, B, D, ncl, noise)  # unseen members of the same clusters
    opos, osc, _ = oracle.topk(q, ref, k)
    in_margin = ((ref @ (q[0] / np.linalg.norm(q[0]))) / np.linalg.norm(ref, axis=1) > osc[0, -1] - 0.0078).sum()
    assert in_margin > 3000, in_margin  # the situation the test is about
    for kernel in ("mfma", "wave"):
        s.set_kernel(kernel)
        ids, sc, _ = s.search_vectors(None, k, q)
        np.testing.assert_array_equal(ids, opos)
        np.testing.assert_allclose(sc, osc.astype(np.float32), atol=1e-7)
        st = s.last_stats()
        assert st["overflow_reruns"] == 0
        # a bf16-only screen would keep the whole cluster (in_margin ~ 10^4 rows per query); at this small N the
        # scan is over after ~3 blocks per wave, so most survivors are early, loose-threshold ones
        assert st["candidates"] < 0.6 * in_margin * B, (st, in_margin)
    s.close()


def test_similarity_matrices(ctx, oracle, g77):
    q, m = g77["queries"], g77["corpus"]
    np.testing.assert_allclose(pa.dot_product(ctx, q, m), g77["dot_f32"], atol=1e-4)
    np.testing.assert_allclose(pa.cosine_similarity_multi_query(ctx, q, m), g77["cos_multi_f32"], atol=TOL)
    np.testing.assert_allclose(pa.cosine_similarity_single_query(ctx, q[0], m), g77["cos_multi_f32"][0], atol=TOL)
    np.testing.assert_allclose(pa.cosine_similarity_multi_query(ctx, q, m), oracle.cosine_similarity_multi_query(q, m), atol=1e-6)


def test_sharded_search_and_merge(ctx, oracle):
    # two shards of one corpus in one process: per-shard device lists -> merge == unsharded oracle
    N, B, k = 30_000, 6, 10
    ref = oracle.synth_rows(77, 0, N, 384)
    q = oracle.synth_rows(78, 0, B, 384)
    cut = 17_000
    hit_bytes = 24
    d_lists = ctx.alloc(2 * B * k * hit_bytes)
    shards = []
    for r, (lo, hi) in enumerate([(0, cut), (cut, N)]):
        s = pa.Searcher(ctx, 384, "cosine")
        s.add_synthetic(1, hi - lo, 77, first_row=lo)
        s.finalize()
        s.set_shard_offset(lo)
        s.search_device(None, k, q, d_lists + r * B * k * hit_bytes)
        shards.append(s)
    ids, scores, counts = pa.merge_topk(ctx, "cosine", 384, d_lists, 2, B, k)
    opos, osc, _ = oracle.topk(q, ref, k)
    np.testing.assert_array_equal(ids, opos)  # synthetic ids = global row index
    np.testing.assert_allclose(scores, osc.astype(np.float32), atol=1e-7)
    raw = ctx.to_host(d_lists, 2 * B * k * hit_bytes).view(np.dtype([("score", "<f8"), ("pos", "<i8"), ("id", "<i8")]))
    assert (raw["pos"][B * k :] >= cut).all() and (raw["pos"][: B * k] < cut).all()
    ctx.free(d_lists)
    for s in shards:
        s.close()


def _two_shard_step(ctx, shards, q, k, dim):
    """The multi-GPU step with both 'ranks' in this process: begin on each shard into one gathered
    buffer (stride B*k+1), flagged merge, end; repeated while any shard reports an overflow."""
    B = q.shape[0]
    rec = B * k + 1
    d_lists = ctx.alloc(len(shards) * rec * 24)
    try:
        for attempt in range(8):
            overs = []
            for r, s in enumerate(shards):
                s.search_device_begin(None, k, q, d_lists + r * rec * 24)
                overs.append(s.search_device_end())  # one stream: collect before the next shard reuses nothing shared
            ids, scores, counts, any_over = pa.merge_topk(ctx, "cosine", dim, d_lists, len(shards), B, k, flagged=True)
            assert any_over == any(overs)
            if not any_over:
                return ids, scores, counts, attempt
        raise AssertionError("still overflowing")
    finally:
        ctx.free(d_lists)


def test_begin_end_protocol_two_shards(ctx, oracle):
    N, B, k = 30_000, 6, 10
    ref = oracle.synth_rows(77, 0, N, 384)
    q = oracle.synth_rows(78, 0, B, 384)
    cut = 13_000
    shards = []
    for lo, hi in [(0, cut), (cut, N)]:
        s = pa.Searcher(ctx, 384, "cosine")
        s.add_synthetic(1, hi - lo, 77, first_row=lo)
        s.finalize()
        s.set_shard_offset(lo)
        shards.append(s)
    ids, scores, counts, attempts = _two_shard_step(ctx, shards, q, k, 384)
    opos, osc, _ = oracle.topk(q, ref, k)
    np.testing.assert_array_equal(ids, opos)
    np.testing.assert_allclose(scores, osc.astype(np.float32), atol=1e-7)
    assert attempts == 0 and (counts == k).all()
    # protocol errors: a second begin without end, an end without begin, more queries than one pass holds
    d = ctx.alloc((200 * k + 1) * 24)
    shards[0].search_device_begin(None, k, q, d)
    with pytest.raises(pa.PcvError):
        shards[0].search_device_begin(None, k, q, d)
    assert shards[0].search_device_end() is False
    with pytest.raises(pa.PcvError):
        shards[0].search_device_end()
    with pytest.raises(pa.PcvError) as ei:
        shards[0].search_device_begin(None, k, oracle.synth_rows(5, 0, 200, 384), d)  # (among ranks a pass is 128 queries, whatever copies a rank holds)
    assert ei.value.status == 3
    ctx.free(d)
    for s in shards:
        s.close()


def test_begin_end_when_a_shard_holds_none_or_many_of_the_selected_sources(ctx, oracle):
    # shard 0: sources 1..11 (11 segments > one launch group); shard 1: source 20 only.  Filtering by source
    # leaves one shard without rows — it must still deliver the same payload layout (begin/end protocol)
    rng = np.random.default_rng(31)
    D, k = 64, 5
    parts0 = [(src, rng.standard_normal((60 + src, D)).astype(np.float32)) for src in range(1, 12)]
    part1 = rng.standard_normal((500, D)).astype(np.float32)
    n0 = sum(p.shape[0] for _, p in parts0)
    s0 = pa.Searcher(ctx, D, "cosine")
    pos = 0
    for src, m in parts0:
        s0.add_rows(src, m, np.arange(pos, pos + m.shape[0]))
        pos += m.shape[0]
    s0.finalize()
    s1 = pa.Searcher(ctx, D, "cosine")
    s1.add_rows(20, part1, np.arange(n0, n0 + 500))
    s1.finalize()
    s1.set_shard_offset(n0)
    q = rng.standard_normal((3, D)).astype(np.float32)
    allm = np.concatenate([m for _, m in parts0] + [part1])
    rec = 3 * k + 1
    d = ctx.alloc(2 * rec * 24)

    def step(sources):
        for r, s in enumerate((s0, s1)):
            s.search_device_begin(sources, k, q, d + r * rec * 24)
            assert s.search_device_end() is False
        ids, scores, counts, over = pa.merge_topk(ctx, "cosine", D, d, 2, 3, k, flagged=True)
        assert over is False
        return ids, counts

    ids, counts = step([20])  # only shard 1 has it
    np.testing.assert_array_equal(ids, oracle.topk(q, part1, k)[0] + n0)
    ids, counts = step(None)  # everything: shard 0 needs two launch groups
    np.testing.assert_array_equal(ids, oracle.topk(q, allm, k)[0])
    ids, counts = step([3])  # only shard 0
    lo = sum(p.shape[0] for src, p in parts0 if src < 3)
    np.testing.assert_array_equal(ids, oracle.topk(q, parts0[2][1], k)[0] + lo)
    ids, counts = step([99])  # nobody
    assert (counts == 0).all() and (ids == -1).all()
    ctx.free(d)
    s0.close()
    s1.close()


def test_begin_end_overflow_is_repeated_by_all_shards(ctx, oracle):
    # adversarial order on shard 1 only (see test_candidate_overflow_reruns_and_stays_exact): its overflow
    # record must reach the merged flag, the step is repeated, and the answer is exact
    rng = np.random.default_rng(9)
    N, D, k = 40_000, 64, 10
    q = rng.standard_normal((1, D)).astype(np.float32)
    noise = rng.standard_normal((N, D)).astype(np.float32)
    t = np.linspace(-1, 1, N, dtype=np.float32)[:, None]
    hard = (t * q + 0.01 * noise).astype(np.float32)
    easy = rng.standard_normal((5_000, D)).astype(np.float32)
    m = np.concatenate([easy, hard])
    shards = []
    for lo, part in [(0, easy), (easy.shape[0], hard)]:
        s = pa.Searcher(ctx, D, "cosine")
        s.add_rows(1, part, np.arange(lo, lo + part.shape[0]))
        s.finalize()
        s.set_shard_offset(lo)
        shards.append(s)
    shards[1].set_candidate_capacity(64)
    ids, scores, counts, attempts = _two_shard_step(ctx, shards, q, k, D)
    opos, osc, _ = oracle.topk(q, m, k)
    np.testing.assert_array_equal(ids, opos)
    assert attempts >= 1
    assert shards[1].last_stats()["overflow_reruns"] == 0  # stats are per call; the rerun was a fresh begin
    for s in shards:
        s.close()


def test_native_rccl_exchange_single_rank(ctx, oracle):
    # pcv_comm_* + pcv_searcher_search_sharded at world=1 (the box has one GPU): RCCL is loaded by the
    # library, the all-gather runs on its stream, and the result equals the plain search and the oracle
    N, B, k = 40_000, 5, 10
    ref = oracle.synth_rows(91, 0, N, 384)
    q = oracle.synth_rows(92, 0, B, 384)
    s = pa.Searcher(ctx, 384, "cosine")
    s.add_synthetic(1, N, 91)
    s.finalize()
    comm = pa.NativeComm(ctx, 1, 0, pa.NativeComm.unique_id())
    for _ in range(2):  # second call reuses the communicator's buffers
        ids, scores, counts = s.search_sharded(comm, None, k, q)
    opos, osc, _ = oracle.topk(q, ref, k)
    np.testing.assert_array_equal(ids, opos)
    np.testing.assert_allclose(scores, osc.astype(np.float32), atol=1e-7)
    assert (counts == k).all()
    ids2, scores2, _ = s.search_vectors(None, k, q)
    np.testing.assert_array_equal(ids, ids2)
    np.testing.assert_array_equal(scores, scores2)
    sh = pa.ShardedSearcher(None, "cosine", 384, searcher=s, ctx=ctx, comm=comm)
    np.testing.assert_array_equal(sh.search_vectors(None, k, q)[0], opos)
    with pytest.raises(pa.PcvError):
        pa.NativeComm(ctx, 2, 2, bytes(128))  # rank outside the world: refused before RCCL is touched
    # more queries than one pass holds -> the sequential form; and an overflowing pass is repeated
    qm = oracle.synth_rows(93, 0, 150, 384)
    np.testing.assert_array_equal(s.search_sharded(comm, None, k, qm)[0], oracle.topk(qm, ref, k)[0])
    s.close()
    rng = np.random.default_rng(9)
    qa = rng.standard_normal((1, 64)).astype(np.float32)
    t = np.linspace(-1, 1, 40_000, dtype=np.float32)[:, None]
    hard = (t * qa + 0.01 * rng.standard_normal((40_000, 64)).astype(np.float32)).astype(np.float32)
    s = build(ctx, hard)
    s.set_candidate_capacity(64)
    np.testing.assert_array_equal(s.search_sharded(comm, None, k, qa)[0], oracle.topk(qa, hard, k)[0])
    assert s.last_stats()["overflow_reruns"] >= 1
    comm.close()
    s.close()


def test_native_sharded_with_filters_and_many_segments(ctx, oracle):
    # pcv_searcher_search_sharded at world=1: a source filter that matches nothing on this shard (the
    # pass is skipped, the exchange still runs and must be waited for), more than 8 segments, an empty filter
    rng = np.random.default_rng(77)
    D, k = 64, 6
    s = pa.Searcher(ctx, D, "cosine")
    parts = []
    for src in range(12):
        m = rng.standard_normal((50 + 11 * src, D)).astype(np.float32)
        parts.append(m)
        s.add_rows(src, m, 1000 * src + np.arange(m.shape[0]))
    s.finalize()
    allm = np.concatenate(parts)
    allids = np.concatenate([1000 * i + np.arange(p.shape[0]) for i, p in enumerate(parts)])
    comm = pa.NativeComm(ctx, 1, 0, pa.NativeComm.unique_id())
    q = rng.standard_normal((5, D)).astype(np.float32)
    for _ in range(3):  # stale pinned results of an earlier call must never be returned
        ids, sc, cnt = s.search_sharded(comm, None, k, q)
        np.testing.assert_array_equal(ids, allids[oracle.topk(q, allm, k)[0]])
        ids, sc, cnt = s.search_sharded(comm, [999], k, q)
        assert (cnt == 0).all() and (ids == -1).all()
        ids, sc, cnt = s.search_sharded(comm, [], k, q)
        assert (cnt == 0).all() and (ids == -1).all()
        ids, sc, cnt = s.search_sharded(comm, [4, 9], k, q)
        sub = np.concatenate([parts[4], parts[9]])
        subids = np.concatenate([4000 + np.arange(parts[4].shape[0]), 9000 + np.arange(parts[9].shape[0])])
        np.testing.assert_array_equal(ids, subids[oracle.topk(q, sub, k)[0]])
    # the device wrappers treat an empty filter like search_vectors does (search.rs:166)
    d = ctx.alloc((5 * k + 1) * 24)
    s.search_device([], k, q, d)
    got = ctx.to_host(d, 5 * k * 24).view(pa.HIT_DTYPE)
    assert (got["pos"] == -1).all()
    s.search_device_begin([], k, q, d)
    assert s.search_device_end() is False
    assert (ctx.to_host(d, 5 * k * 24).view(pa.HIT_DTYPE)["pos"] == -1).all()
    ctx.free(d)
    # concurrent plain and sharded searches on one searcher serialise on its mutex
    import threading

    errs = []

    def plain():
        try:
            for _ in range(20):
                got, _, _ = s.search_vectors(None, k, q)
                np.testing.assert_array_equal(got, allids[oracle.topk(q, allm, k)[0]])
        except Exception as e:  # pragma: no cover
            errs.append(e)

    t = threading.Thread(target=plain)
    t.start()
    for _ in range(20):
        ids, _, _ = s.search_sharded(comm, None, k, q)
        np.testing.assert_array_equal(ids, allids[oracle.topk(q, allm, k)[0]])
    t.join()
    assert not errs, errs
    comm.close()
    s.close()


def test_process_exit_with_live_handles_is_clean():
    # nothing closed, objects die in whatever order the interpreter picks; and a context closed before
    # its searcher / model: the context takes its handles down first, exit code 0 either way
    import subprocess
    import sys

    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import perceive_amd as pa\n"
        "ctx = pa.Context(0); s = pa.Searcher(ctx, 64, 'cosine'); s.add_synthetic(1, 1000, 3); s.finalize()\n"
        "m = pa.Model(ctx, pa.make_desc(300, 128, 1, 4, 256, 64), synthetic_seed=1)\n"
        "c = pa.NativeComm(ctx, 1, 0, pa.NativeComm.unique_id())\n"
        "print(s.search_vectors(None, 3, np.ones((1, 64), np.float32))[0][0][0])\n"
        "MODE\n"
    ) % ROOT
    for mode in ("pass", "ctx.close(); s.close(); m.close(); c.close()"):
        r = subprocess.run([sys.executable, "-c", code.replace("MODE", mode)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (mode, r.returncode, r.stderr[-800:])


def test_error_paths(ctx):
    s = pa.Searcher(ctx, 16, "cosine")
    s.add_rows(1, np.ones((4, 16), np.float32))
    with pytest.raises(pa.PcvError) as e:
        s.search_vectors(None, 3, np.ones((1, 16), np.float32))
    assert "finalize" in str(e.value)
    s.finalize()
    with pytest.raises(pa.PcvError):
        s.search_vectors(None, 0, np.ones((1, 16), np.float32))
    ids, _, counts = s.search_vectors(None, 1000, np.ones((1, 16), np.float32))  # more than one pass ranks: fine (4 rows, 4 hits)
    assert counts[0] == 4 and sorted(ids[0][:4]) == [0, 1, 2, 3] and (ids[0][4:] == -1).all()
    with pytest.raises(pa.PcvError):
        s.search_vectors(None, (1 << 24) + 1, np.ones((1, 16), np.float32))
    with pytest.raises(ValueError):
        s.search_vectors(None, 3, np.ones((1, 8), np.float32))
    s.close()
    with pytest.raises(pa.PcvError):
        pa.Searcher(ctx, 0, "cosine")


def test_dim_768_and_more_than_64_queries(ctx, oracle):
    # the product's default model (MsMarcoBertBaseDotV5) is 768-d and ranks by dot product;
    # 130 queries = three passes of <= 64
    rng = np.random.default_rng(31)
    N, D, B = 20_000, 768, 130
    m = rng.standard_normal((N, D)).astype(np.float32)
    q = rng.standard_normal((B, D)).astype(np.float32)
    for metric in ("dot", "cosine"):
        s = build(ctx, m, metric=metric)
        ids, sc, cnt = s.search_vectors(None, 10, q)
        opos, osc, _ = oracle.topk(q, m, 10, metric=1 if metric == "dot" else 0)
        np.testing.assert_array_equal(ids, opos)
        if metric == "cosine":
            np.testing.assert_allclose(sc, osc.astype(np.float32), atol=1e-7)
        else:
            np.testing.assert_allclose(sc, np.maximum(0, 1 - osc / D).astype(np.float32), atol=1e-6)
        s.close()


def test_concurrent_searches_from_threads(ctx, oracle):
    # `&self` search is called from a thread pool in the reference (Arc<Searcher>, app_state.rs:52-57)
    import threading

    rng = np.random.default_rng(41)
    m = rng.standard_normal((30_000, 384)).astype(np.float32)
    s = build(ctx, m)
    qs = rng.standard_normal((8, 3, 384)).astype(np.float32)
    exp = [oracle.topk(qs[i], m, 5)[0] for i in range(8)]
    got, errs = [None] * 8, []

    def work(i):
        try:
            for _ in range(5):
                got[i] = s.search_vectors(None, 5, qs[i])[0]
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    for i in range(8):
        np.testing.assert_array_equal(got[i], exp[i])
    s.close()


_TWO_CONTEXTS = r"""
import sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_ffi
import perceive_amd as pa
orc = oracle_ffi.load()
rng = np.random.default_rng(7)
m = rng.standard_normal((6000, 384)).astype(np.float32)
wide = rng.standard_normal((600, 8192)).astype(np.float32)
q = rng.standard_normal((200, 384)).astype(np.float32)     # 200 queries: the 102 KB tile of scan_mfma8_hold_kernel<3,4>
qw = rng.standard_normal((4, 8192)).astype(np.float32)     # 4 x 8192-d: 128 KB of queries in scan_wave_kernel
exp, expw = orc.topk(q, m, 10)[0], orc.topk(qw, wide, 5)[0]
go = threading.Barrier(2)
out, errs = {}, []
def work(i):
    try:
        ctx = pa.Context(0)                                   # a context of its own per thread (same device)
        s = pa.Searcher(ctx, 384, "cosine"); s.add_rows(1, m); s.finalize()
        w = pa.Searcher(ctx, 8192, "cosine"); w.add_rows(1, wide); w.finalize()
        go.wait()                                             # both threads launch each kernel for the first time together
        a = s.search_vectors(None, 10, q)[0]
        go.wait()
        b = w.search_vectors(None, 5, qw)[0]
        out[i] = (a, b, s.last_stats()["screening_copy"], w.last_stats()["kernel_used"])
        s.close(); w.close(); ctx.close()
    except Exception as e:
        errs.append(repr(e))
        try: go.abort()
        except Exception: pass
th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
assert not errs, errs
for i in range(2):
    a, b, copy, kern = out[i]
    assert copy == 2 and kern == 1, (copy, kern)
    assert (a == exp).all() and (b == expw).all()
print("two contexts ok")
"""


def test_two_contexts_two_threads_first_use_of_large_lds_kernels():
    """Kernels that need more than 64 KB of dynamic LDS are allowed that per device, under a lock, at first use
    (common.h: allow_dynamic_lds) — not through unsynchronised function statics.  In a fresh process (no kernel has run yet)
    two threads, each with a context and searchers of its own, launch the 102 KB-tile int8 scan and the 128 KB wave scan
    for the first time at the same moment."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TWO_CONTEXTS, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "two contexts ok" in r.stdout, r.stdout + r.stderr


def test_create_destroy_does_not_leak(ctx):
    import ctypes as C

    from perceive_amd import _ffi

    hip = C.CDLL("libamdhip64.so")
    free0, total = C.c_size_t(), C.c_size_t()
    rng = np.random.default_rng(1)
    m = rng.standard_normal((50_000, 384)).astype(np.float32)
    q = rng.standard_normal((5, 384)).astype(np.float32)

    def cycle():
        s = build(ctx, m)
        s.search_vectors(None, 10, q)
        s.close()

    cycle()
    ctx.synchronize()
    hip.hipMemGetInfo(C.byref(free0), C.byref(total))
    for _ in range(5):
        cycle()
    ctx.synchronize()
    free1 = C.c_size_t()
    hip.hipMemGetInfo(C.byref(free1), C.byref(total))
    assert free0.value - free1.value < 64 << 20, (free0.value, free1.value)


@pytest.mark.parametrize("D", [1, 3, 50, 65, 200, 1000])
def test_awkward_dimensions(ctx, oracle, D):
    # any dim works: features are zero-padded to a multiple of 64 in HBM
    rng = np.random.default_rng(D)
    m = rng.standard_normal((777, D)).astype(np.float32)
    q = rng.standard_normal((7, D)).astype(np.float32)
    for kernel in ("wave", "mfma"):
        s = build(ctx, m, kernel=kernel)
        ids, sc, cnt = s.search_vectors(None, 9, q)
        opos, osc, ocnt = oracle.topk(q, m, 9)
        if D == 1:  # cosine of scalars is +-1: everything ties, order = position among equal signs
            np.testing.assert_allclose(sc, osc.astype(np.float32), atol=1e-7)
        np.testing.assert_array_equal(ids, opos)
        rows, _ = s.get_rows(np.arange(5))
        np.testing.assert_array_equal(rows, m[:5])
        s.close()


def test_large_dim_falls_back_to_wave_kernel(ctx, oracle):
    # 4096-d: the bf16 query tile no longer fits the LDS -> auto picks the wave kernel, forcing mfma fails loudly
    rng = np.random.default_rng(5)
    m = rng.standard_normal((300, 4096)).astype(np.float32)
    q = rng.standard_normal((6, 4096)).astype(np.float32)
    s = build(ctx, m)
    ids, sc, _ = s.search_vectors(None, 5, q)
    np.testing.assert_array_equal(ids, oracle.topk(q, m, 5)[0])
    assert s.last_stats()["kernel_used"] == 1
    s.set_kernel("mfma")
    with pytest.raises(pa.PcvError) as e:
        s.search_vectors(None, 5, q)
    assert e.value.status == 3
    s.close()


def test_empty_and_degenerate_indexes(ctx):
    s = pa.Searcher(ctx, 32, "cosine")
    s.finalize()  # nothing added
    q = np.ones((2, 32), np.float32)
    ids, sc, cnt = s.search_vectors(None, 5, q)
    assert cnt.sum() == 0 and (ids == -1).all() and np.isnan(sc).all()
    assert s.num_rows == 0 and s.source_ids == []
    s.add_rows(3, np.zeros((0, 32), np.float32))  # empty batch creates nothing searchable
    s.finalize()
    s.finalize()  # idempotent
    assert s.search_vectors(None, 5, q)[2].sum() == 0
    s.add_rows(3, np.eye(32, dtype=np.float32)[:1])  # one row, k larger than the corpus
    s.finalize()
    ids, sc, cnt = s.search_vectors([3], 5, q)
    assert cnt.tolist() == [1, 1] and ids[:, 0].tolist() == [0, 0]
    np.testing.assert_allclose(sc[:, 0], 1 / np.sqrt(32), atol=1e-7)
    from perceive_amd import _ffi
    _ffi.check(_ffi.lib().pcv_searcher_clear_source(s._handle, 12345))  # unknown source: no-op (search.rs:58-79)
    s.finalize()
    assert s.num_rows == 1
    s.close()
    s.close()  # double close is harmless


def test_incremental_adds_make_several_segments(ctx, oracle):
    # add -> finalize -> add -> finalize on one source: two segments, one logical source
    rng = np.random.default_rng(77)
    a = rng.standard_normal((1000, 96)).astype(np.float32)
    b = rng.standard_normal((500, 96)).astype(np.float32)
    s = pa.Searcher(ctx, 96, "cosine")
    s.add_rows(1, a, np.arange(1000))
    s.finalize()
    s.add_rows(1, b, 5000 + np.arange(500))
    s.finalize()
    assert s.num_rows == 1500
    q = rng.standard_normal((9, 96)).astype(np.float32)
    ids, sc, _ = s.search_vectors([1], 10, q)
    allm = np.concatenate([a, b])
    allids = np.concatenate([np.arange(1000), 5000 + np.arange(500)])
    np.testing.assert_array_equal(ids, allids[oracle.topk(q, allm, 10)[0]])
    s.close()


def test_screening_copy_is_invisible_in_the_results(ctx, oracle):
    """The bf16 screening copy changes what the coarse screen reads, never what comes back: same ids and bit-equal
    scores with the copy on and off, on Gaussian and on clustered rows, for every batch shape of the MFMA kernel;
    rows added later, cleared sources and source filters keep it in step."""
    n, D = 300_000, 384
    for clusters in (0, 40):
        off = pa.Searcher(ctx, D, "cosine")
        off.set_screening_copy("off")
        off.add_synthetic(1, n, 0xC0FFEE, n_clusters=clusters, noise=0.01 if clusters else 0.0)
        off.finalize()
        rng = np.random.default_rng(3 + clusters)
        probe = off.get_rows(rng.integers(0, n, 128))[0]
        q = (probe + 0.3 * rng.standard_normal(probe.shape)).astype(np.float32) if clusters else rng.standard_normal((128, D)).astype(np.float32)
        opos, _, _ = oracle.topk(q[:3], off.get_rows(np.arange(n))[0], 10)
        for mode, code in (("int8", 2), ("bf16", 1)):
            on = pa.Searcher(ctx, D, "cosine")
            on.set_screening_copy(mode)
            on.add_synthetic(1, n, 0xC0FFEE, n_clusters=clusters, noise=0.01 if clusters else 0.0)
            on.finalize()
            for B in (1, 4, 5, 33, 64, 128):
                a = on.search_vectors(None, 10, q[:B])
                assert on.last_stats()["screening_copy"] == code and on.last_stats()["kernel_used"] == 2
                b = off.search_vectors(None, 10, q[:B])
                assert off.last_stats()["screening_copy"] == 0
                np.testing.assert_array_equal(a[0], b[0])
                np.testing.assert_array_equal(a[1], b[1])
            np.testing.assert_array_equal(on.search_vectors(None, 10, q[:3])[0], opos)
            on.close()
        off.close()
    # incremental adds, a second source, clearing: the copy follows at every finalize
    rng = np.random.default_rng(9)
    base = rng.standard_normal((5000, 128)).astype(np.float32)
    s = pa.Searcher(ctx, 128, "cosine")
    s.add_rows(1, base, np.arange(5000))
    s.finalize()
    assert s.search_vectors(None, 1, base[77:78])[0][0, 0] == 77 and s.last_stats()["screening_copy"] == 2  # auto = int8
    extra = rng.standard_normal((333, 128)).astype(np.float32)  # lands in the spare room of the same segment, mid-block
    s.add_rows(1, extra, 5000 + np.arange(333))
    s.add_rows(2, -extra, 9000 + np.arange(333))
    s.finalize()
    got = s.search_vectors(None, 1, extra[10:11])
    assert got[0][0, 0] == 5010 and abs(got[1][0, 0] - 1.0) < 1e-6 and s.last_stats()["screening_copy"] == 2
    assert s.search_vectors([2], 1, -extra[5:6])[0][0, 0] == 9005 and s.last_stats()["screening_copy"] == 2
    everything = np.concatenate([base, extra, -extra])
    qs = rng.standard_normal((7, 128)).astype(np.float32)
    ref_ids = np.concatenate([np.arange(5333), 9000 + np.arange(333)])[oracle.topk(qs, everything, 10)[0]]
    np.testing.assert_array_equal(s.search_vectors(None, 10, qs)[0], ref_ids)
    s.set_screening_copy("off")  # frees the copies at once
    np.testing.assert_array_equal(s.search_vectors(None, 10, qs)[0], ref_ids)
    assert s.last_stats()["screening_copy"] == 0
    s.set_screening_copy("bf16")  # built by the next finalize
    np.testing.assert_array_equal(s.search_vectors(None, 10, qs)[0], ref_ids)
    assert s.last_stats()["screening_copy"] == 0
    s.finalize()
    np.testing.assert_array_equal(s.search_vectors(None, 10, qs)[0], ref_ids)
    assert s.last_stats()["screening_copy"] == 1
    s.set_screening_copy("auto")  # the other kind replaces it at the next finalize
    s.finalize()
    np.testing.assert_array_equal(s.search_vectors(None, 10, qs)[0], ref_ids)
    assert s.last_stats()["screening_copy"] == 2
    s.rebuild_source([], 1)  # clears source 1
    assert s.search_vectors(None, 1, -extra[5:6])[0][0, 0] == 9005 and s.num_rows == 333
    assert s.last_stats()["screening_copy"] == 2
    # dot metric: the copy holds the rows themselves (scale 1)
    d = pa.Searcher(ctx, 128, "dot")
    d.add_rows(1, base, np.arange(5000))
    d.finalize()
    dd = pa.Searcher(ctx, 128, "dot")
    dd.set_screening_copy("off")
    dd.add_rows(1, base, np.arange(5000))
    dd.finalize()
    a, b = d.search_vectors(None, 10, qs), dd.search_vectors(None, 10, qs)
    assert d.last_stats()["screening_copy"] == 2 and dd.last_stats()["screening_copy"] == 0
    db = pa.Searcher(ctx, 128, "dot")
    db.set_screening_copy("bf16")
    db.add_rows(1, base, np.arange(5000))
    db.finalize()
    c_ = db.search_vectors(None, 10, qs)
    assert db.last_stats()["screening_copy"] == 1
    np.testing.assert_array_equal(c_[0], b[0])
    np.testing.assert_array_equal(c_[1], b[1])
    db.close()
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for x in (s, d, dd):
        x.close()


@pytest.mark.parametrize("D", [384, 320, 192, 96])  # 320, 192: padded width 64 mod 128 (half-filled last swizzle group of the query tile)
@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_int8_screen_on_rows_that_quantise_badly(ctx, oracle, D, metric):
    """The int8 screen's margin is certified per (row, query) from the quantisation steps: rows it represents badly may
    only cost extra fine-screen work, never a wrong answer.  Heavy-tailed rows, one-hot rows, thousands of rows within
    one quantisation step of the query, norms across 30 decades (cosine) and within a factor 1e3 (dot)."""
    rng = np.random.default_rng(D + len(metric))
    n = 6000
    heavy = rng.standard_cauchy((n, D)).astype(np.float32)                     # one feature dominates each row
    onehot = np.zeros((500, D), np.float32)
    onehot[np.arange(500), rng.integers(0, D, 500)] = rng.standard_normal(500).astype(np.float32)
    q0 = rng.standard_normal(D).astype(np.float32)
    near = (q0[None, :] * (1.0 + 1e-4 * rng.standard_normal((3000, 1))) + 2e-4 * rng.standard_normal((3000, D))).astype(np.float32)
    gauss = rng.standard_normal((n, D)).astype(np.float32)
    corpus = np.concatenate([heavy, onehot, near, gauss])
    if metric == "cosine":
        corpus *= (10.0 ** rng.uniform(-15, 15, (corpus.shape[0], 1))).astype(np.float32)
    else:
        corpus *= (10.0 ** rng.uniform(-1.5, 1.5, (corpus.shape[0], 1))).astype(np.float32)
    corpus = corpus[rng.permutation(corpus.shape[0])]
    queries = np.concatenate([q0[None, :], rng.standard_normal((5, D)).astype(np.float32), rng.standard_cauchy((3, D)).astype(np.float32),
                              np.eye(D, dtype=np.float32)[:2], corpus[:3]])
    # the same queries, perturbed, as batches of 100 and 256: the block-holding forms of the int8 scan
    many = (queries[np.arange(256) % len(queries)] * (1.0 + 0.05 * rng.standard_normal((256, D)))).astype(np.float32)
    for mode, code in (("int8", 2), ("bf16", 1), ("off", 0)):
        s = build(ctx, corpus, metric=metric, screen=mode, kernel="mfma")
        for qs in (queries[:1], queries) + ((many[:100], many) if mode == "int8" else ()):
            ids, scores, counts = s.search_vectors(None, 10, qs)
            assert s.last_stats()["screening_copy"] == code
            opos, osc, _ = oracle.topk(qs, corpus, 10, metric=1 if metric == "dot" else 0)
            np.testing.assert_array_equal(ids, opos)
        s.close()


def test_int8_screen_leaves_wide_rows_to_the_other_paths(ctx, oracle):
    # beyond 1024 padded features the integer dot product no longer converts to f32 exactly: AUTO keeps the bf16 copy for such
    # corpora, and asking for the int8 form is an error at finalize
    rng = np.random.default_rng(11)
    m = rng.standard_normal((400, 1100)).astype(np.float32)
    q = rng.standard_normal((6, 1100)).astype(np.float32)
    s = build(ctx, m, kernel="mfma")
    ids, _, _ = s.search_vectors(None, 5, q)
    np.testing.assert_array_equal(ids, oracle.topk(q, m, 5)[0])
    assert s.last_stats()["kernel_used"] == 2 and s.last_stats()["screening_copy"] == 1
    s.close()
    e = pa.Searcher(ctx, 1100, "cosine")
    e.set_screening_copy("int8")
    e.add_rows(1, m)
    with pytest.raises(pa.PcvError) as err:
        e.finalize()
    assert err.value.status == 3
    e.close()


def test_screening_copy_gives_way_when_memory_is_short(ctx, oracle, monkeypatch):
    """AUTO: a failed allocation of a screening copy switches the copies off for the searcher and the f32 rows are scanned;
    asked for explicitly, the same failure is an error at finalize."""
    rng = np.random.default_rng(21)
    m = rng.standard_normal((3000, 128)).astype(np.float32)
    q = rng.standard_normal((9, 128)).astype(np.float32)
    ref = oracle.topk(q, m, 10)[0]
    s = pa.Searcher(ctx, 128, "cosine")
    s.add_rows(1, m[:2000], np.arange(2000))
    s.finalize()
    np.testing.assert_array_equal(s.search_vectors(None, 10, q)[0], oracle.topk(q, m[:2000], 10)[0])
    assert s.last_stats()["screening_copy"] == 2
    s.set_tuning(fail_copy_alloc=True)
    s.add_rows(2, m[2000:], 2000 + np.arange(1000))  # a second source = a new segment, whose copy cannot be allocated
    s.finalize()
    np.testing.assert_array_equal(s.search_vectors(None, 10, q)[0], ref)
    assert s.last_stats()["screening_copy"] == 0 and s.last_stats()["kernel_used"] == 2  # every copy was given back
    s.set_tuning()
    s.finalize()  # still off for this searcher: it gave way for good
    np.testing.assert_array_equal(s.search_vectors(None, 10, q)[0], ref)
    assert s.last_stats()["screening_copy"] == 0
    s.set_screening_copy("auto")  # asked again: rebuilt at the next finalize
    s.finalize()
    np.testing.assert_array_equal(s.search_vectors(None, 10, q)[0], ref)
    assert s.last_stats()["screening_copy"] == 2
    # ... and so does the staged rebuild of a source (Searcher::rebuild_source): the old and the new rows of the source are
    # resident at once; if AUTO gave its copies up to fit them, the old rows going is the room to have them again
    s.set_tuning(fail_copy_alloc=True)
    s.add_rows(3, m[:500], 5000 + np.arange(500))
    s.finalize()
    s.set_tuning()
    assert s.search_vectors(None, 10, q) is not None and s.last_stats()["screening_copy"] == 0  # gave way again
    s.rebuild_source([(2000 + i, 2, m[2000 + i]) for i in range(1000)] + [(9, 7, m[0])], 2)  # (rows of other sources are ignored)
    got = s.search_vectors([1, 2], 10, q)[0]
    np.testing.assert_array_equal(got, ref)
    assert s.last_stats()["screening_copy"] == 2  # the copies are back without being asked for
    s.close()
    e = pa.Searcher(ctx, 128, "cosine")
    e.set_tuning(fail_copy_alloc=True)
    e.set_screening_copy("int8")
    e.add_rows(1, m)
    with pytest.raises(pa.PcvError):
        e.finalize()
    e.close()


@pytest.mark.parametrize("D,B,k", [(64, 1, 1), (100, 7, 10), (256, 33, 128), (512, 65, 10), (768, 128, 10), (1000, 5, 3), (1024, 64, 10), (384, 128, 128),
                                   (96, 100, 10), (256, 128, 10), (320, 65, 5),   # the block-holding form for 65..128 queries
                                   (384, 256, 10), (200, 130, 7), (96, 300, 3),    # ... and for 129..256 in one pass
                                   (640, 33, 10), (896, 64, 5), (640, 3, 4)])      # 5 / 7 chunks per block: the drain form's run-time chunk count
@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_screens_agree_with_the_oracle_across_shapes(ctx, oracle, D, B, k, metric):
    """int8 / bf16 / no screening copy over widths, batch sizes and k that exercise every query-tile shape, three sources in
    five segments (rows added in two rounds), planted duplicates (ties) and a zero row: ids equal to the oracle's."""
    rng = np.random.default_rng(D * 1000 + B)
    n = 4000
    m = rng.standard_normal((n, D)).astype(np.float32)
    if metric == "dot":
        m *= rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    m[1234] = m[77]          # exact duplicate: the lower position wins
    m[2500] = 0.0            # no score under cosine; score 0 under dot
    q = rng.standard_normal((B, D)).astype(np.float32)
    q[0] = m[77]
    ids = np.arange(n, dtype=np.int64) + 10_000
    src = np.where(np.arange(n) < 1500, 1, np.where(np.arange(n) < 3100, 2, 3))
    ref = oracle.topk(q, m, k, metric=1 if metric == "dot" else 0)[0]
    results = {}
    for mode in ("int8", "bf16", "off"):
        s = pa.Searcher(ctx, D, metric)
        s.set_screening_copy(mode)
        s.set_kernel("mfma")
        for lo, hi in ((0, 900), (1500, 2000), (3100, 4000)):       # first round
            s.add_rows(int(src[lo]), m[lo:hi], ids[lo:hi])
        s.finalize()
        for lo, hi in ((900, 1500), (2000, 3100)):                  # second round, after a finalize
            s.add_rows(int(src[lo]), m[lo:hi], ids[lo:hi])
        s.finalize()
        got, sc, cnt = s.search_vectors(None, k, q)
        assert s.last_stats()["screening_copy"] == {"int8": 2, "bf16": 1, "off": 0}[mode]
        results[mode] = (got, sc)
        s.close()
    # rows of a source are not contiguous in insertion order: compare as the oracle ranks them, by id
    pos_of_id = {int(i): p for p, i in enumerate(ids)}
    for mode, (got, sc) in results.items():
        np.testing.assert_array_equal(sc, results["off"][1])
        np.testing.assert_array_equal(got, results["off"][0])
    got = results["int8"][0]
    # the searcher's global positions follow source order, so ties between equal scores may resolve differently from the
    # oracle's row order: compare scores of the returned rows and the sets where scores are distinct
    for b in range(B):
        valid = ref[b] >= 0
        assert (got[b] >= 0).sum() == valid.sum()
        mine = np.array([oracle.canonical_score(q[b], m[pos_of_id[int(i)]], metric=1 if metric == "dot" else 0) for i in got[b] if i >= 0])
        theirs = np.array([oracle.canonical_score(q[b], m[int(p)], metric=1 if metric == "dot" else 0) for p in ref[b] if p >= 0])
        np.testing.assert_array_equal(mine, theirs)


def test_speculative_threshold_is_checked_and_repeated_when_it_fails(ctx, oracle, monkeypatch):
    # The int8 scan starts from a guess taken from the seed rows (512 blocks of 32 rows spread evenly over the first
    # segment) and checked at the end of the pass (scan.h).  Here the guess must fail for query 0: its ten best rows
    # ARE seed rows, one in each of the ten seed groups (position in the sample mod k), so the seed score the guess is taken from has
    # fewer than ten rows at or above it; the pass is repeated without the guess and the answer is the oracle's.
    # Query 1 has its best rows outside the seed blocks: its guess holds.  Afterwards the same searcher answers
    # unrelated queries without a repeat, and a searcher with the guess switched off (PCV_SCAN_FLAGS bit 5) returns the
    # same hits.
    rng = np.random.default_rng(77)
    N, D, k = 400_000, 128, 10
    m = rng.standard_normal((N, D)).astype(np.float32)
    q = rng.standard_normal((2, D)).astype(np.float32)
    nblocks, shift = (N + 31) // 32, 0
    while (512 << (shift + 1)) <= nblocks:
        shift += 1
    seed_rows = [((7 + 31 * j) << shift) * 32 + j for j in range(k)]          # row j of seed block 7 + 31 j: ten seed groups
    other_rows = [(((11 + 29 * j) << shift) + 1) * 32 + 5 for j in range(k)]  # the block after a seed block
    assert len({((7 + 31 * j) * 32 + j) % k for j in range(k)}) == k and max(seed_rows + other_rows) < N  # (position in the sample) mod k
    for j in range(k):  # graded near-copies of the queries
        m[seed_rows[j]] = q[0] + (0.02 + 0.01 * j) * rng.standard_normal(D).astype(np.float32)
        m[other_rows[j]] = q[1] + (0.02 + 0.01 * j) * rng.standard_normal(D).astype(np.float32)
    s = build(ctx, m, kernel="mfma")
    ids, scores, counts = s.search_vectors(None, k, q)
    st = s.last_stats()
    assert st["screening_copy"] == 2 and st["speculation_reruns"] == 1 and st["scan_launches"] == 2, st
    opos, osc, _ = oracle.topk(q, m, k)
    np.testing.assert_array_equal(ids, opos)
    np.testing.assert_allclose(scores, osc, rtol=0, atol=1e-6)
    assert set(ids[0]) == set(seed_rows) and set(ids[1]) == set(other_rows)
    q2 = rng.standard_normal((64, D)).astype(np.float32)
    for _ in range(6):  # enough passes for the learned part of the guess to come into play
        ids2, sc2, _ = s.search_vectors(None, k, q2)
    assert s.last_stats()["speculation_reruns"] == 0
    opos2, osc2, _ = oracle.topk(q2, m, k)
    np.testing.assert_array_equal(ids2, opos2)
    s.close()
    monkeypatch.setenv("PCV_SCAN_FLAGS", "32")
    s = build(ctx, m, kernel="mfma")
    ids3, sc3, _ = s.search_vectors(None, k, q)
    assert s.last_stats()["speculation_reruns"] == 0 and s.last_stats()["scan_launches"] == 1
    np.testing.assert_array_equal(ids3, ids)
    np.testing.assert_array_equal(sc3, scores)
    s.close()


@pytest.mark.parametrize("metric,k", [("cosine", 10), ("dot", 2), ("cosine", 128)])
def test_speculative_threshold_on_rows_sorted_by_similarity(ctx, oracle, metric, k):
    # Rows stored best first: block 0 — a seed block — holds the 32 best rows for the query, so every guess taken from
    # the seed rows is too high; and the reverse order, where the seed rows say nothing about the end of the corpus.
    # Both must come out exact, whatever the number of repeated passes.
    rng = np.random.default_rng(5)
    N, D = 300_000, 96
    q = rng.standard_normal((1, D)).astype(np.float32)
    qh = (q / np.linalg.norm(q)).astype(np.float64)
    u = rng.standard_normal((N, D))
    u -= (u @ qh.T) * qh  # orthogonal to the query, unit length: similarity is a function of t alone
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    t = np.linspace(1.0, -1.0, N)[:, None]
    m = (t * qh + 0.5 * u).astype(np.float32)
    for order in (m, m[::-1].copy()):
        s = build(ctx, order, metric=metric, kernel="mfma")
        s.set_candidate_capacity(1 << 16)
        ids, scores, counts = s.search_vectors(None, k, q)
        st = s.last_stats()
        opos, osc, _ = oracle.topk(q, order, k, metric=1 if metric == "dot" else 0)
        np.testing.assert_array_equal(ids, opos)
        assert st["screening_copy"] == 2 and st["scan_launches"] == 1 + st["overflow_reruns"] + st["speculation_reruns"], st
        if order is m and k >= 10:  # (k = 2: no seed slot is a safe enough guess on 300 000 rows, none is made)
            assert st["speculation_reruns"] == 1, st
        s.close()


@pytest.mark.parametrize("kind", ["gauss", "clustered", "heavy_tails", "duplicates"])
def test_guess_on_and_off_give_the_same_hits(ctx, monkeypatch, kind):
    # the speculative start threshold changes which rows are screened out early, never the answer: the same queries on
    # the same rows with it (default) and without it (PCV_SCAN_FLAGS bit 5) — Gaussian rows, clusters, heavy-tailed
    # features and a corpus of 50-fold duplicates, cosine and dot, 1..200 queries, k from 1 to 100
    rng = np.random.default_rng({"gauss": 1, "clustered": 2, "heavy_tails": 3, "duplicates": 4}[kind])
    n, d = 300_000, 96
    if kind == "gauss":
        m = rng.standard_normal((n, d)).astype(np.float32)
    elif kind == "clustered":
        c = rng.standard_normal((n // 5000, d)).astype(np.float32)
        m = (c[rng.integers(0, len(c), n)] + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
    elif kind == "heavy_tails":
        m = rng.standard_t(2.5, (n, d)).astype(np.float32)
    else:
        base = rng.standard_normal((n // 50, d)).astype(np.float32)
        m = base[rng.integers(0, len(base), n)].copy()
    shapes = ((1, 10), (7, 1), (64, 10), (128, 100), (64, 10), (64, 10), (200, 2))
    queries = [(m[rng.integers(0, n, B)] + 0.3 * rng.standard_normal((B, d))).astype(np.float32) for B, _ in shapes]
    for metric in ("cosine", "dot"):
        hits = {}
        for flags in ("0", "32"):
            monkeypatch.setenv("PCV_SCAN_FLAGS", flags)
            s = build(ctx, m, metric=metric, kernel="mfma")
            hits[flags] = [s.search_vectors(None, k, q)[:2] for (B, k), q in zip(shapes, queries)]
            s.close()
        for (ia, sa), (ib, sb) in zip(hits["0"], hits["32"]):
            np.testing.assert_array_equal(ia, ib)
            np.testing.assert_array_equal(sa, sb)


def _clustered(oracle, n, d, n_clusters, noise, seed, nq, rng):
    rows = oracle.synth_rows_clustered(seed, 0, n, d, n_clusters, noise)
    probe = rows[rng.integers(0, n, nq)]
    return rows, (probe + 0.5 * noise * rng.standard_normal(probe.shape)).astype(np.float32)


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_mid_copy_is_invisible_in_the_results_and_cuts_the_f32_reads(ctx, oracle, metric):
    """The row-major 16-bit mid copy (scan.h): on a clustered corpus the coarse int8 screen lets a whole cluster through per
    query; with the mid copy nearly all of those are ruled out from 2 bytes per feature and only a few rows have their f32 row
    read.  Hits and scores are the oracle's with the copy forced on, off, and built by AUTO after two such passes; rows added
    later join the copy; a failed allocation switches AUTO off for good and makes ON an error."""
    rng = np.random.default_rng(31)
    n, d, k = 240_000, 128, 10
    rows, q = _clustered(oracle, n, d, 16, 0.004, 0xC1, 64, rng)
    ref = oracle.topk(q, rows, k, 1 if metric == "dot" else 0)
    s = pa.Searcher(ctx, d, metric)
    s.add_rows(1, rows[:200_000], np.arange(200_000))
    s.finalize()
    sub = oracle.topk(q, rows[:200_000], k, 1 if metric == "dot" else 0)

    def check(ids, sc, want):
        np.testing.assert_array_equal(ids, want[0])
        if metric == "cosine":
            np.testing.assert_allclose(sc, want[1].astype(np.float32), rtol=0, atol=1e-6)

    s.set_mid_copy("off")
    ids, sc, _ = s.search_vectors(None, k, q)
    st0 = s.last_stats()
    check(ids, sc, sub)
    assert st0["screening_copy"] == 2 and st0["mid_copy"] == 0 and st0["coarse_survivors"] > 4096 * 64, st0
    s.set_mid_copy("auto")
    for i in range(2):  # two passes above the trigger: the third call queues the build of the copy, beside the searches
        ids, sc, _ = s.search_vectors(None, k, q)
        assert s.last_stats()["mid_copy"] == 0
        check(ids, sc, sub)
    for i in range(200):  # ... and the calls go on without the copy until it is there
        ids, sc, _ = s.search_vectors(None, k, q)
        st1 = s.last_stats()
        check(ids, sc, sub)
        if st1["mid_copy"] == 1:
            break
    assert st1["mid_copy"] == 1 and 0 < st1["mid_survivors"] < st1["coarse_survivors"] // 8, (i, st1)
    # rows added afterwards join the copy at finalize (AUTO keeps what it has built)
    s.add_rows(1, rows[200_000:], np.arange(200_000, n))
    s.finalize()
    ids, sc, _ = s.search_vectors(None, k, q)
    check(ids, sc, ref)
    assert s.last_stats()["mid_copy"] == 1
    # a second source whose copies cannot be allocated: AUTO gives the mid copy (and the int8 copies) up for good, results unchanged
    extra = (rows[:1000] * 0.5).astype(np.float32)
    s.set_tuning(fail_copy_alloc=True)
    s.add_rows(2, extra, n + np.arange(1000))
    s.finalize()
    s.set_tuning()
    ids, sc, _ = s.search_vectors([1], k, q)
    check(ids, sc, ref)
    assert s.last_stats()["mid_copy"] == 0 and s.last_stats()["screening_copy"] == 0
    s.close()
    # ON: built at finalize, also on a corpus the trigger would never see; OFF afterwards frees it
    t = pa.Searcher(ctx, d, metric)
    t.set_mid_copy("on")
    t.add_rows(7, rows[:50_000], np.arange(50_000))
    t.finalize()
    ids, sc, _ = t.search_vectors(None, k, q[:5])
    assert t.last_stats()["mid_copy"] == 1
    want = oracle.topk(q[:5], rows[:50_000], k, 1 if metric == "dot" else 0)
    check(ids, sc, want)
    t.set_mid_copy("off")
    ids2, sc2, _ = t.search_vectors(None, k, q[:5])
    assert t.last_stats()["mid_copy"] == 0
    np.testing.assert_array_equal(ids, ids2)
    np.testing.assert_array_equal(sc, sc2)
    t.close()
    u = pa.Searcher(ctx, d, metric)
    u.set_mid_copy("on")
    u.set_tuning(fail_copy_alloc=True)
    u.set_screening_copy("off")
    u.add_rows(1, rows[:1000])
    with pytest.raises(pa.PcvError):
        u.finalize()
    u.close()


def test_auto_mid_copy_is_built_beside_the_searches(ctx):
    """AUTO decides inside a search call that the mid copy pays (two passes in a row with a crowd at the coarse screen).  The
    build — a pass over every row, several times what a search takes — is queued on a stream of its own: the deciding call and
    the calls after it take what a call without the copy takes (round 3 built it inside the deciding call: 228 ms on a
    100M-row corpus), and once the build's event has come the passes use it.  Same hits throughout."""
    import time

    n, d, B, k = 8_000_000, 384, 64, 10
    s = pa.Searcher(ctx, d, "cosine")
    s.add_synthetic(1, n, 0x5EED, n_clusters=n // 20_000, noise=0.004)
    s.finalize()
    rng = np.random.default_rng(5)
    probe = s.get_rows(rng.integers(0, n, B))[0]
    q = (probe + 0.002 * rng.standard_normal(probe.shape)).astype(np.float32)
    s.set_mid_copy("off")
    times = []
    for _ in range(6):
        t0 = time.perf_counter()
        ref = s.search_vectors(None, k, q)[0]
        times.append(time.perf_counter() - t0)
    base = float(np.median(times[2:]))
    assert s.last_stats()["coarse_survivors"] > 4096 * B  # the AUTO trigger's first condition
    s.set_mid_copy("auto")
    calls, seen = [], None
    for i in range(400):
        t0 = time.perf_counter()
        ids = s.search_vectors(None, k, q)[0]
        calls.append(time.perf_counter() - t0)
        np.testing.assert_array_equal(ids, ref)
        if s.last_stats()["mid_copy"] == 1:
            seen = i
            break
    assert seen is not None and seen >= 2, seen  # (calls 0 and 1 are the two hot passes; call 2 queues the build)
    # No call waits for the build.  The deciding call pays for queueing it (a stream, a helper thread) and for the moments the
    # helper thread's allocations hold the runtime's lock (measured: 8.9 ms against 3.3 ms here; 32 ms against 5.8 ms at 50M
    # rows, where the build itself is ~20 ms of kernel time and the older kernel inside the call was > 110 ms); the calls
    # while it runs share the memory system with it.
    assert max(calls[: seen + 1]) < 4.0 * base + 1e-3, (base, calls[: seen + 1])
    assert seen <= 8, calls
    t0 = time.perf_counter()
    ids = s.search_vectors(None, k, q)[0]
    with_copy = time.perf_counter() - t0
    np.testing.assert_array_equal(ids, ref)
    assert with_copy < base  # what the copy is for on such rows
    s.close()


def test_mid_copy_on_rows_that_quantise_badly(ctx, oracle):
    """Rows with one dominant feature (the per-row 16-bit scale is set by it, everything else lands on a few levels), zero rows,
    rows of tiny norm, a zero query: the mid screen's bound must hold for each; forced on, against the oracle."""
    rng = np.random.default_rng(32)
    n, d, k = 40_000, 96, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[::7, 3] *= 400.0
    rows[5::11] *= 1e-6
    rows[100] = 0.0
    q = rng.standard_normal((33, d)).astype(np.float32)
    q[4] = rows[7] * 2.0
    q[9] = 0.0
    for metric in ("cosine", "dot"):
        s = pa.Searcher(ctx, d, metric)
        s.set_mid_copy("on")
        s.add_rows(1, rows)
        s.finalize()
        ids, sc, cnt = s.search_vectors(None, k, q)
        assert s.last_stats()["mid_copy"] == 1 and s.last_stats()["screening_copy"] == 2
        opos, osc, ocnt = oracle.topk(q, rows, k, 1 if metric == "dot" else 0)
        np.testing.assert_array_equal(cnt, ocnt)
        np.testing.assert_array_equal(ids, opos)
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [520, 768, 1024])
@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_mid_copy_on_rows_wider_than_512_features(ctx, oracle, metric, dim):
    """A mid row wider than 512 features is more than one 16-byte piece per lane: the mid screen has to sum all of them
    (a build that read 64 pieces only scored the first 512 features, dropped rows of the top-k and let the thresholds
    stand still: 11 700 coarse survivors per query instead of 1 100 on the 768-d bench corpus).  Rows of unequal norm as in
    the product's default model (768-d, dot metric); forced on, against the oracle, and against the pass without the copy."""
    rng = np.random.default_rng(dim)
    n, k = 30_000, 10
    rows = rng.standard_normal((n, dim)).astype(np.float32) * rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    rows[:, dim - 40:] *= 3.0  # weight in the features past the first 512: a partial sum would rank differently
    q = rng.standard_normal((40, dim)).astype(np.float32)
    q[:, dim - 40:] *= 3.0
    s = pa.Searcher(ctx, dim, metric)
    s.set_mid_copy("on")
    s.add_rows(1, rows)
    s.finalize()
    ids, sc, cnt = s.search_vectors(None, k, q)
    st = s.last_stats()
    assert st["mid_copy"] == 1 and st["screening_copy"] == 2
    opos, osc, ocnt = oracle.topk(q, rows, k, 1 if metric == "dot" else 0)
    np.testing.assert_array_equal(cnt, ocnt)
    np.testing.assert_array_equal(ids, opos)
    s.set_mid_copy("off")
    ids2, _, _ = s.search_vectors(None, k, q)
    st2 = s.last_stats()
    np.testing.assert_array_equal(ids2, opos)
    # the thresholds rise as fast with the copy as without it: about as many rows pass the coarse screen
    assert st["coarse_survivors"] < 2 * st2["coarse_survivors"] + 1000, (st["coarse_survivors"], st2["coarse_survivors"])
    s.close()


@pytest.mark.gpu
def test_mid_copy_is_used_only_by_passes_that_stream_the_int8_copy(ctx, oracle):
    """The mid screen's bound needs |q'|_1 of the pass's queries, which only the int8 path computes.  A pass that streams the
    bf16 copy or the f32 rows must not take the mid rows (it used to, with whatever constants the last int8 pass had left:
    here that pass has one-hot queries, |q'|_1 = 1, fifteen times smaller than that of the dense queries that follow).
    Near-tie data: many rows within 1e-5 of each query's k-th best."""
    rng = np.random.default_rng(77)
    n, d, k = 60_000, 384, 10
    base = rng.standard_normal((n // 200, d)).astype(np.float32)
    rows = (np.repeat(base, 200, axis=0) + 2e-5 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    q = (base[:48] + 1e-5 * rng.standard_normal((48, d)).astype(np.float32)).astype(np.float32)
    onehot = np.zeros((48, d), np.float32)
    onehot[np.arange(48), np.arange(48)] = 1.0
    opos, osc, ocnt = oracle.topk(q, rows, k, 0)
    s = pa.Searcher(ctx, d, "cosine")
    s.set_mid_copy("on")
    s.add_rows(1, rows)
    s.finalize()
    s.search_vectors(None, k, onehot)  # an int8 pass with the mid copy: leaves |q'|_1 = 1 behind
    assert s.last_stats()["mid_copy"] == 1 and s.last_stats()["screening_copy"] == 2
    ids, sc, cnt = s.search_vectors(None, k, q)  # int8 + mid copy, its own constants
    np.testing.assert_array_equal(ids, opos)
    for mode, copy in (("bf16", 1), ("off", 0)):
        s.search_vectors(None, k, onehot) if mode == "bf16" else None
        s.set_screening_copy(mode)
        s.finalize()
        ids, sc, cnt = s.search_vectors(None, k, q)
        st = s.last_stats()
        assert st["screening_copy"] == copy and st["mid_copy"] == 0 and st["mid_survivors"] == 0, st
        np.testing.assert_array_equal(cnt, ocnt)
        np.testing.assert_array_equal(ids, opos)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,B", [(384, 100), (384, 256), (256, 200), (768, 100)])
def test_mid_copy_under_the_block_holding_and_128_query_forms(ctx, oracle, dim, B):
    """The mid screen sits in the fine screen all int8 scans share; the tests above reach it from the 64-query tile kernel.
    Here: the block-holding form at 128 and 256 queries (384-d, 256-d) and the eight-wave 128-query tile (768-d), on a
    clustered corpus where most coarse survivors end at the mid screen, against the oracle and against the pass without the copy."""
    rng = np.random.default_rng(dim + B)
    n, k = 96_000, 10
    rows, q = _clustered(oracle, n, dim, 12, 0.004, 0xC7, B, rng)
    opos, osc, ocnt = oracle.topk(q, rows, k, 0)
    s = pa.Searcher(ctx, dim, "cosine")
    s.set_mid_copy("on")
    s.add_rows(1, rows)
    s.finalize()
    ids, sc, cnt = s.search_vectors(None, k, q)
    st = s.last_stats()
    assert st["mid_copy"] == 1 and st["screening_copy"] == 2, st  # (a pass may be repeated: lists that overflow on 8 000-row clusters)
    assert 0 < st["mid_survivors"] < st["coarse_survivors"] // 4, st
    np.testing.assert_array_equal(cnt, ocnt)
    np.testing.assert_array_equal(ids, opos)
    np.testing.assert_allclose(sc, osc.astype(np.float32), rtol=0, atol=1e-6)
    s.set_mid_copy("off")
    ids2, _, _ = s.search_vectors(None, k, q)
    assert s.last_stats()["mid_copy"] == 0
    np.testing.assert_array_equal(ids2, opos)
    s.close()
