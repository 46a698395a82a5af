"""CPU: the oracle (oracle/scan.c, synth.c) against the committed PyTorch-CPU vectors
(tests/golden/gen_scan_golden.py) and known answers."""
import os
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def g1000(golden_dir):
    return np.load(os.path.join(golden_dir, "scan_n1000_d384.npz"))


@pytest.fixture(scope="module")
def g77(golden_dir):
    return np.load(os.path.join(golden_dir, "scan_n77_d100.npz"))


def test_dot_product_matches_torch(oracle, g1000):
    out = oracle.dot_product(g1000["queries"], g1000["corpus"])
    np.testing.assert_allclose(out, g1000["dot_f32"], rtol=0, atol=1e-3)  # |dot| ~ 20, f32 ordering noise


def test_cosine_multi_query_matches_torch(oracle, g1000):
    out = oracle.cosine_similarity_multi_query(g1000["queries"], g1000["corpus"])
    ref = g1000["cos_multi_f32"]
    # zero row 500: no epsilon in lib.rs:74-75 -> NaN on both sides
    assert np.isnan(out[:, 500]).all() and np.isnan(ref[:, 500]).all()
    keep = np.ones(ref.shape[1], bool)
    keep[500] = False
    np.testing.assert_allclose(out[:, keep], ref[:, keep], rtol=0, atol=1e-4)  # north_star tolerance
    assert np.abs(out[:, keep] - ref[:, keep]).max() < 5e-6


def test_cosine_single_query_matches_torch(oracle, g1000):
    out = oracle.cosine_similarity_single_query(g1000["queries"][0], g1000["corpus"])
    ref = g1000["cos_single_q0_f32"]
    keep = ~np.isnan(ref)
    assert keep.sum() == 999
    np.testing.assert_allclose(out[keep], ref[keep], rtol=0, atol=1e-4)


def test_odd_shape_matches_torch(oracle, g77):
    out = oracle.cosine_similarity_multi_query(g77["queries"], g77["corpus"])
    np.testing.assert_allclose(out, g77["cos_multi_f32"], rtol=0, atol=1e-4)
    pos, sc, cnt = oracle.topk(g77["queries"], g77["corpus"], int(g77["k"]))
    assert (cnt == int(g77["k"])).all()
    np.testing.assert_array_equal(pos, g77["topk_f64"])


def test_canonical_topk_matches_torch_f64(oracle, g1000):
    k = int(g1000["k"])
    pos, sc, cnt = oracle.topk(g1000["queries"], g1000["corpus"], k)
    assert (cnt == k).all()
    np.testing.assert_array_equal(pos, g1000["topk_f64"])
    ref = np.take_along_axis(g1000["cos_f64"], g1000["topk_f64"], 1)
    np.testing.assert_allclose(sc, ref, rtol=0, atol=1e-12)
    # and against the f32 reference-shaped ranking wherever the rank-k gap is not an f32 tie
    clear = g1000["rank_gap_f64"] > 1e-5
    assert clear.sum() >= 60
    for b in np.nonzero(clear)[0]:
        assert set(pos[b]) == set(g1000["topk_f32"][b])


def test_ties_break_to_lower_position(oracle, g1000):
    # query 5 is a near copy of row 123; rows 777 (duplicate) and 778 (3x scaled) have the same cosine
    pos, sc, _ = oracle.topk(g1000["queries"][5:6], g1000["corpus"], 3)
    assert set(pos[0]) == {123, 777, 778}
    p = list(pos[0])
    assert p.index(123) < p.index(777)  # exact duplicate: identical f64 score, lower position first
    assert abs(sc[0][2] - sc[0][0]) < 1e-9  # the 3x copy differs only by the f32 rounding of 3*x
    assert 500 not in oracle.topk(g1000["queries"][:4], g1000["corpus"], 1000)[0]  # zero row never returned


def test_topk_fewer_valid_rows_than_k(oracle):
    m = np.zeros((5, 8), np.float32)
    m[1, 0] = 1.0
    m[3, 1] = 2.0
    q = np.array([[1.0, 1.0, 0, 0, 0, 0, 0, 0]], np.float32)
    pos, sc, cnt = oracle.topk(q, m, 4)
    assert cnt[0] == 2 and list(pos[0]) == [1, 3, -1, -1]
    np.testing.assert_allclose(sc[0][:2], [2**-0.5, 2**-0.5])


def test_ndarray_distance_and_search_vector(oracle):
    rng = np.random.default_rng(1)
    m = rng.standard_normal((200, 16)).astype(np.float32)
    q = rng.standard_normal(16).astype(np.float32)
    d = np.array([oracle.ndarray_distance(q, m[i]) for i in range(200)])
    ref = np.maximum(0.0, 1.0 - (m @ q) / 16.0)
    np.testing.assert_allclose(d, ref, atol=1e-6)
    ids = np.arange(1000, 1200)
    sor = np.where(np.arange(200) < 120, 7, 9)
    out_ids, out_d = oracle.search_vector(q, m, ids, sor, [9], 5)
    mask = sor == 9
    order = np.argsort(-(m[mask].astype(np.float64) @ q.astype(np.float64)), kind="stable")[:5]
    np.testing.assert_array_equal(out_ids, ids[mask][order])
    assert (np.diff(out_d) >= 0).all()  # ascending distance, search.rs:179
    np.testing.assert_allclose(out_d, ref[mask][order], atol=1e-6)
    # clamp at 0 (search.rs:277)
    big = (q * 100).astype(np.float32)
    assert oracle.ndarray_distance(q, big) == 0.0


def test_blob_codec_known_answers(oracle):
    v = np.array([1.0, -2.5, 0.0, 3.4028235e38, 1e-45], np.float32)
    blob = oracle.serialize_embedding(v)
    assert blob == struct.pack("<5f", *v)
    assert blob[:4] == bytes([0x00, 0x00, 0x80, 0x3F])  # 1.0f little-endian
    np.testing.assert_array_equal(oracle.deserialize_embedding(blob), v)
    assert len(oracle.serialize_embedding(np.zeros(384, np.float32))) == 1536


def test_philox_known_answer(oracle):
    import ctypes as C

    # Random123 known-answer vectors for philox4x32-10
    def ph(ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        oracle.lib.orc_philox4x32_10(c, k, o)
        return [hex(x) for x in o]

    assert ph([0, 0, 0, 0], [0, 0]) == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert ph([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert ph([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        "0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_synth_rows_statistics_and_determinism(oracle):
    a = oracle.synth_rows(0x5EED, 0, 2000, 384)
    b = oracle.synth_rows(0x5EED, 1000, 10, 384)
    np.testing.assert_array_equal(a[1000:1010], b)  # counter-based: rows independent of batch
    assert abs(a.mean()) < 0.01 and abs(a.std() - 1.0) < 0.01
    n = oracle.synth_rows(0x5EED, 0, 16, 384, normalize=True)
    np.testing.assert_allclose(np.linalg.norm(n.astype(np.float64), axis=1), 1.0, atol=1e-6)
    assert not np.array_equal(oracle.synth_rows(1, 0, 4, 384), oracle.synth_rows(2, 0, 4, 384))


def test_cpu_baseline_agrees_with_oracle(oracle, g1000):
    k = int(g1000["k"])
    for shaped in (False, True):
        secs, pos, sc = oracle.baseline_scan(g1000["queries"][:8], g1000["corpus"], k, threads=2, shaped=shaped)
        assert secs > 0
        clear = g1000["rank_gap_f64"][:8] > 1e-5
        for b in np.nonzero(clear)[0]:
            assert set(pos[b]) == set(g1000["topk_f64"][b])
        ref = np.take_along_axis(g1000["cos_f64"][:8], pos, 1)
        np.testing.assert_allclose(sc, ref, atol=1e-5)
