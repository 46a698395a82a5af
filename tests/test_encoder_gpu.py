"""GPU parity tests of the encoder (pcv_model_*), through the C ABI, against the CPU oracle
(oracle/encoder.c) and the committed Hugging Face BertModel vectors.

Tolerance: the scan's parity bar is cosine within 1e-4 (BASELINE.json north_star); embeddings are
unit vectors, so 1e-4 absolute per component is the matching bar here.  f32 end to end on both
sides: the observed differences are ~1e-6 (summation order only)."""
import os

import numpy as np
import pytest

import perceive_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
TOL = 1e-4


def load_tiny(golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_tiny.npz"))
    v, h, ly, nh, it, mp = [int(x) for x in g["desc"]]
    desc = dict(vocab=v, hidden=h, layers=ly, heads=nh, inter=it, max_pos=mp, eps=1e-12, pooling=0, normalize=1)
    weights = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    return g, desc, weights


def make_model(ctx, desc, weights=None, seed=0, **kw):
    d = pa.make_desc(desc["vocab"], desc["hidden"], desc["layers"], desc["heads"], desc["inter"], desc["max_pos"],
                     layer_norm_eps=desc.get("eps", 1e-12), **kw)
    m = pa.Model(ctx, d, synthetic_seed=seed)
    if weights is not None:
        m.load_state_dict(weights)
    return m


def test_hf_golden_tiny(ctx, oracle, golden_dir):
    g, desc, weights = load_tiny(golden_dir)
    m = make_model(ctx, desc, weights)
    out = m.encode_tokens(g["ids"], g["mask"])
    assert np.abs(out - g["normed"]).max() < TOL
    B, L = g["ids"].shape
    msk = g["mask"].astype(bool)
    for ly in range(desc["layers"] + 1):
        hid = m.debug_hidden(ly, B, L)
        assert np.abs(hid[msk] - g["hidden"][ly][msk]).max() < TOL, ly
    # and the C oracle on the same inputs (padded positions too: same -10000 mask convention)
    oout, ohid = oracle.encode_tokens(desc, weights, g["ids"], g["mask"], want_hidden=True)
    np.testing.assert_allclose(out, oout, atol=1e-5)
    np.testing.assert_allclose(m.debug_hidden(desc["layers"], B, L), ohid[-1], atol=5e-5)
    m.close()
    # un-normalised mean pooling (e.g. msmarco dot-product models: has_normalization() false, model.rs:151)
    m2 = make_model(ctx, desc, weights, normalize=False)
    assert np.abs(m2.encode_tokens(g["ids"], g["mask"]) - g["mean"]).max() < TOL
    m2.close()


def test_ragged_lengths_and_padding_invariance(ctx, oracle, golden_dir):
    g, desc, weights = load_tiny(golden_dir)
    m = make_model(ctx, desc, weights)
    rng = np.random.default_rng(0)
    # L not a multiple of 32, one-token and full-length rows, batch of 7
    toks = [list(rng.integers(1, desc["vocab"], n)) for n in (1, 5, 33, 40, 17, 2, 40)]
    ids, mask = m.generate_token_tensors(toks)
    assert ids.shape == (7, 40) and mask.sum(1).tolist() == [1, 5, 33, 40, 17, 2, 40]
    out = m.encode_tokens(ids, mask)
    oout, _ = oracle.encode_tokens(desc, weights, ids, mask)
    np.testing.assert_allclose(out, oout, atol=2e-5)
    # each row alone (no padding) gives the same embedding: tokenize.rs pads to the batch max
    for i in (0, 2, 4):
        ids1, mask1 = m.generate_token_tensors([toks[i]])
        np.testing.assert_allclose(m.encode_tokens(ids1, mask1)[0], out[i], atol=2e-5)
    m.close()


@pytest.mark.parametrize("pooling", ["cls", "max", "mean_sqrt_len"])
def test_pooling_modes(ctx, oracle, golden_dir, pooling):
    g, desc, weights = load_tiny(golden_dir)
    code = {"cls": 1, "max": 2, "mean_sqrt_len": 3}[pooling]
    m = make_model(ctx, desc, weights, pooling=pooling, normalize=False)
    out = m.encode_tokens(g["ids"], g["mask"])
    oout, _ = oracle.encode_tokens(dict(desc, pooling=code, normalize=0), weights, g["ids"], g["mask"])
    np.testing.assert_allclose(out, oout, atol=5e-5)
    m.close()


def test_dense_module(ctx, oracle, golden_dir):
    g, desc, weights = load_tiny(golden_dir)
    rng = np.random.default_rng(1)
    w2 = dict(weights)
    w2["dense.linear.weight"] = (rng.standard_normal((64, desc["hidden"])) * 0.1).astype(np.float32)
    w2["dense.linear.bias"] = (rng.standard_normal(64) * 0.1).astype(np.float32)
    m = make_model(ctx, desc, w2, dense_out=64, dense_activation="tanh", normalize=True)
    assert m.output_dim == 64
    out = m.encode_tokens(g["ids"], g["mask"])
    oout, _ = oracle.encode_tokens(dict(desc, dense_out=64, dense_act=1, normalize=1), w2, g["ids"], g["mask"])
    np.testing.assert_allclose(out, oout, atol=2e-5)
    m.close()


def test_minilm_shape_synthetic_weights(ctx, oracle):
    # the BASELINE model shape (384 hidden, 6 layers, 12 heads x 32, FFN 1536) with the library's
    # seeded synthetic weights, downloaded and fed to the oracle
    m = pa.Model(ctx, synthetic_seed=7)
    assert m.output_dim == 384 and m.model_type.model_id == 0
    sd = m.state_dict()
    assert sd["embeddings.word_embeddings.weight"].shape == (30522, 384)
    rng = np.random.default_rng(2)
    toks = [list(rng.integers(1000, 30000, n)) for n in (48, 7, 31, 20)]
    ids, mask = m.generate_token_tensors(toks)
    out = m.encode_tokens(ids, mask)
    desc = dict(vocab=30522, hidden=384, layers=6, heads=12, inter=1536, max_pos=512, eps=1e-12, pooling=0, normalize=1)
    oout, ohid = oracle.encode_tokens(desc, sd, ids, mask, want_hidden=True)
    assert np.abs(out - oout).max() < TOL
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
    assert np.abs(m.debug_hidden(6, 4, 48) - ohid[-1]).max() < 5e-4  # activations are O(1..3)
    st = m.last_stats()
    assert st["batch"] == 4 and st["seq_len"] == 48 and st["flops"] > 0 and st["total_ms"] > 0
    # cosine ranking from GPU embeddings == from oracle embeddings (the end-to-end contract)
    assert (np.argsort(-(out @ out.T), 1) == np.argsort(-(oout @ oout.T), 1)).all()
    m.close()


@pytest.mark.parametrize("shape", [(32, 256), (40, 200), (9, 100), (97, 250)])
def test_minilm_shape_few_tokens_32_row_layernorm_tiles(ctx, oracle, shape):
    # One rank's share of BASELINE configs[4] (32 documents x 256 tokens = 8 192 tokens) and other token counts that fill the
    # chip better in 32-row tiles than in 64-row ones: the projection + LayerNorm fusion runs on 32-row tiles whose eight waves
    # split K and stage their operands privately (gemm_f32_ln32_kernel); (40, 200) ends in a partial tile, (9, 100) is a handful
    # of tiles, (97, 250) = 24 250 tokens are 758 tiles: three (or two) per workgroup, the last one partial.  At these token counts
    # the QKV projection runs on 128 x 96 tiles (gemm_f32_n96_kernel; (40, 200) and (97, 250) end in a partial row tile).  Sampled documents against the C oracle (padding invariance makes
    # single-row oracle runs valid), and the whole batch against the 64-row form of the same fusion (PCV_NO_LN32 is read once
    # per process, so the comparison model is the all-in-one-batch run of the documents one by one).
    B, L = shape
    m = pa.Model(ctx, synthetic_seed=7)
    rng = np.random.default_rng(B * 1000 + L)
    lens = rng.integers(max(8, L // 8), L + 1, B)
    lens[0] = L
    toks = [list(rng.integers(1000, 30000, int(n))) for n in lens]
    ids, mask = m.generate_token_tensors(toks)
    assert ids.shape == (B, L)
    emb = m.encode_tokens(ids, mask)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    desc = dict(vocab=30522, hidden=384, layers=6, heads=12, inter=1536, max_pos=512, eps=1e-12, pooling=0, normalize=1)
    sd = m.state_dict()
    for i in sorted({0, 1, B // 2, B - 1}):
        ids1, mask1 = m.generate_token_tensors([toks[i]])
        oout, ohid = oracle.encode_tokens(desc, sd, ids1, mask1, want_hidden=True)
        assert np.abs(emb[i] - oout[0]).max() < TOL, i
    # the last layer's hidden states of document 0 (full length): every token row of a tile, not only the pooled vector
    if B * L <= 16384:  # (hidden states are kept for debugging up to that many tokens)
        ids1, mask1 = m.generate_token_tensors([toks[0]])
        _, ohid = oracle.encode_tokens(desc, sd, ids1, mask1, want_hidden=True)
        assert np.abs(m.debug_hidden(6, B, L)[0] - ohid[-1][0]).max() < 5e-4
    else:  # more sampled documents instead, from every third of the batch (a workgroup's first, second and third tile)
        for i in range(5, B, 13):
            ids1, mask1 = m.generate_token_tensors([toks[i]])
            oout, _ = oracle.encode_tokens(desc, sd, ids1, mask1)
            assert np.abs(emb[i] - oout[0]).max() < TOL, i
    m.close()


def test_head_dim_64_and_long_sequence(ctx, oracle):
    # bert-base-like heads (64 wide) and L = 300 > one 128-key chunk: exercises the online softmax
    desc = dict(vocab=500, hidden=256, layers=1, heads=4, inter=512, max_pos=512, eps=1e-12, pooling=0, normalize=1)
    m = make_model(ctx, desc, seed=3)
    sd = m.state_dict()
    rng = np.random.default_rng(4)
    toks = [list(rng.integers(1, 500, n)) for n in (300, 129, 64)]
    ids, mask = m.generate_token_tensors(toks)
    out = m.encode_tokens(ids, mask)
    oout, _ = oracle.encode_tokens(desc, sd, ids, mask)
    assert np.abs(out - oout).max() < TOL
    m.close()


def test_weight_file_roundtrip_and_errors(ctx, golden_dir, tmp_path):
    g, desc, weights = load_tiny(golden_dir)
    path = str(tmp_path / "tiny.pcvw")
    pa.save_weights(path, weights)
    d = pa.make_desc(desc["vocab"], desc["hidden"], desc["layers"], desc["heads"], desc["inter"], desc["max_pos"])
    m = pa.Model(ctx, d, weights_path=path)
    assert np.abs(m.encode_tokens(g["ids"], g["mask"]) - g["normed"]).max() < TOL
    np.testing.assert_array_equal(m.get_tensor("encoder.layer.1.output.dense.bias"), weights["encoder.layer.1.output.dense.bias"])
    with pytest.raises(pa.ModelError):
        m.encode_tokens(np.ones((1, 100), np.int64), np.ones((1, 100), np.int64))  # L > max_position_embeddings
    with pytest.raises(pa.ModelError):
        m.encode(["needs a tokenizer"])
    m.close()
    with pytest.raises(pa.PcvError):
        pa.Model(ctx, d, weights_path=str(tmp_path / "missing.pcvw"))
    bad = pa.make_desc(100, 100, 1, 4, 256, 64)  # hidden not a multiple of 128
    with pytest.raises(pa.PcvError) as e:
        pa.Model(ctx, bad)
    assert e.value.status == 3


@pytest.mark.parametrize("compute", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("shape", ["tiny", "minilm"])
def test_split_precision_modes_are_f32_accurate(ctx, oracle, golden_dir, shape, compute):
    # PCV_COMPUTE_BF16X3: operands split into three bf16 terms, six bf16 MFMAs per product.
    # PCV_COMPUTE_F16X2 : two f16 terms (power-of-two rescaled), three f16 MFMAs per product.
    # Same bar as the exact-f32 mode (1e-4 on unit-norm embeddings); observed error stays at the 1e-6 level.
    if shape == "tiny":
        g, desc, weights = load_tiny(golden_dir)
        m = make_model(ctx, desc, weights, compute=compute)
        out = m.encode_tokens(g["ids"], g["mask"])
        assert np.abs(out - g["normed"]).max() < TOL
        oout, _ = oracle.encode_tokens(desc, weights, g["ids"], g["mask"])
        assert np.abs(out - oout).max() < 2e-5
        # weights replaced after creation are re-split
        w2 = {k: (v * 1.5 if k.endswith("intermediate.dense.weight") else v) for k, v in weights.items()}
        m.load_state_dict(w2)
        o2, _ = oracle.encode_tokens(desc, w2, g["ids"], g["mask"])
        assert np.abs(m.encode_tokens(g["ids"], g["mask"]) - o2).max() < 2e-5
        m.close()
    else:
        m = pa.Model(ctx, pa.minilm_l6_desc(compute), synthetic_seed=7)
        mf = pa.Model(ctx, pa.minilm_l6_desc("f32"), synthetic_seed=7)
        rng = np.random.default_rng(2)
        toks = [list(rng.integers(1000, 30000, n)) for n in (48, 7, 31, 20, 64)]
        ids, mask = m.generate_token_tensors(toks)
        a, b = m.encode_tokens(ids, mask), mf.encode_tokens(ids, mask)
        assert np.abs(a - b).max() < 2e-5
        desc = dict(vocab=30522, hidden=384, layers=6, heads=12, inter=1536, max_pos=512, eps=1e-12, pooling=0, normalize=1)
        oout, _ = oracle.encode_tokens(desc, m.state_dict(), ids, mask)
        assert np.abs(a - oout).max() < TOL
        m.close()
        mf.close()


@pytest.mark.parametrize("compute", ["f32", "f16x2"])
@pytest.mark.parametrize("heads,L", [(4, 77), (8, 40), (4, 200), (8, 129), (4, 256), (8, 250), (4, 250), (4, 225)])
def test_attention_shapes_head_dim_64_and_32_ragged(ctx, oracle, compute, heads, L):
    # hidden 256: head_dim 64 (4 heads) and 32 (8 heads); sequence lengths that are not multiples of 32,
    # more than one 128-key chunk, ragged masks — both attention kernels (exact f32 and two-term f16)
    desc = dict(vocab=400, hidden=256, layers=2, heads=heads, inter=512, max_pos=256, eps=1e-12, pooling=0, normalize=1)
    m = make_model(ctx, desc, seed=11, compute=compute)
    rng = np.random.default_rng(heads * 1000 + L)
    B = 5
    ids = rng.integers(1, 400, (B, L)).astype(np.int64)
    mask = np.ones_like(ids)
    for bb, n in enumerate(rng.integers(1, L + 1, B)):
        mask[bb, n:] = 0
    mask[0, :] = 1
    ids *= mask
    out = m.encode_tokens(ids, mask)
    oout, _ = oracle.encode_tokens(desc, m.state_dict(), ids, mask)
    assert np.abs(out - oout).max() < 2e-5
    m.close()


def test_wave_specialised_bf16x3_gemm_is_bit_identical(ctx, tmp_path):
    # PCV_GEMM_WS=1 selects the persistent producer/consumer form of the bf16x3 GEMM: same staging maps,
    # fragment layout and product order, hence the same bits (the switch is read once per process)
    import subprocess
    import sys

    rng = np.random.default_rng(4)
    ids = rng.integers(1000, 30000, (6, 160)).astype(np.int64)  # 960 tokens: 8 row tiles, the last one partial
    mask = np.ones_like(ids)
    mask[2, 100:] = 0
    m = pa.Model(ctx, pa.minilm_l6_desc("bf16x3"), synthetic_seed=3)
    ref = m.encode_tokens(ids * mask, mask)
    m.close()
    np.save(tmp_path / "ids.npy", ids * mask)
    np.save(tmp_path / "mask.npy", mask)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import perceive_amd as pa\n"
        "ctx = pa.Context(0); m = pa.Model(ctx, pa.minilm_l6_desc('bf16x3'), synthetic_seed=3)\n"
        "np.save(%r, m.encode_tokens(np.load(%r), np.load(%r)))\n"
        "m.close(); ctx.close()\n"
    ) % (ROOT, str(tmp_path / "out.npy"), str(tmp_path / "ids.npy"), str(tmp_path / "mask.npy"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PCV_GEMM_WS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    np.testing.assert_array_equal(np.load(tmp_path / "out.npy"), ref)


def test_persistent_attention_walks_several_items_per_workgroup(ctx, oracle):
    # head_dim 64 at 225..256 keys runs the persistent attention kernel: (batch, head) items walked by at most one
    # workgroup per CU.  72 sequences x 4 heads = 288 items > 256 CUs: some workgroups take two items (the next item's
    # K / V pieces in flight under the first one's MFMAs), ragged masks, a length that is not a multiple of 32.
    desc = dict(vocab=400, hidden=256, layers=1, heads=4, inter=512, max_pos=256, eps=1e-12, pooling=0, normalize=1)
    m = make_model(ctx, desc, seed=17)
    rng = np.random.default_rng(99)
    B, L = 72, 250
    ids = rng.integers(1, 400, (B, L)).astype(np.int64)
    mask = np.ones_like(ids)
    for bb, n in enumerate(rng.integers(1, L + 1, B)):
        mask[bb, n:] = 0
    mask[0, :] = 1
    ids *= mask
    out = m.encode_tokens(ids, mask)
    oout, _ = oracle.encode_tokens(desc, m.state_dict(), ids, mask)
    assert np.abs(out - oout).max() < 2e-5
    m.close()


@pytest.mark.parametrize("compute", ["f32", "bf16x3", "f16x2"])
def test_bert_base_width_cls_pooling_no_normalisation(ctx, oracle, compute):
    # the reference's default model (MsMarcoBertBaseDotV5) is BERT-base wide: hidden 768, 12 heads of 64, FFN 3072,
    # CLS pooling, dot-product embeddings (no L2 normalisation).  Two layers of that width against the oracle.
    desc = dict(vocab=1000, hidden=768, layers=2, heads=12, inter=3072, max_pos=128, eps=1e-12, pooling=1, normalize=0)
    m = make_model(ctx, desc, seed=13, compute=compute, pooling="cls", normalize=False)
    rng = np.random.default_rng(5)
    ids = rng.integers(1, 1000, (3, 70)).astype(np.int64)
    mask = np.ones_like(ids)
    mask[1, 33:] = 0
    ids *= mask
    out = m.encode_tokens(ids, mask)
    oout, _ = oracle.encode_tokens(desc, m.state_dict(), ids, mask)
    scale = np.abs(oout).max()
    assert np.abs(out - oout).max() < 2e-5 * max(1.0, scale), (np.abs(out - oout).max(), scale)
    m.close()


def test_f16x2_reports_activation_overflow(ctx, golden_dir):
    # an FFN bias of 1e4 makes the FFN2 input exceed 65504 / 16: the f16 split turns it into inf; the
    # library says so instead of returning NaN embeddings (the f32 mode handles the same model)
    g, desc, weights = load_tiny(golden_dir)
    w = dict(weights)
    k = next(n for n in w if n.endswith("intermediate.dense.bias"))
    w[k] = np.full_like(w[k], 1e4)
    m = make_model(ctx, desc, w, compute="f16x2")
    with pytest.raises(pa.ModelError) as e:
        m.encode_tokens(g["ids"], g["mask"])
    assert "non-finite" in str(e.value)
    m.close()
    mf = make_model(ctx, desc, w, compute="f32")
    assert np.isfinite(mf.encode_tokens(g["ids"], g["mask"])).all()
    mf.close()


def test_f16x2_refuses_weights_outside_f16_range(ctx, golden_dir):
    g, desc, weights = load_tiny(golden_dir)
    w = dict(weights)
    k = next(n for n in w if n.endswith("intermediate.dense.weight"))
    w[k] = w[k].copy()
    w[k].flat[3] = 300.0  # * 2^8 does not fit f16
    m = make_model(ctx, desc, weights, compute="f16x2")
    m.load_state_dict(w)
    with pytest.raises(pa.ModelError) as e:  # ModelError::ModelPanic of the reference
        m.encode_tokens(g["ids"], g["mask"])
    assert "status 3" in str(e.value) and "F16X2" in str(e.value)
    m.load_state_dict(weights)  # back in range: usable again
    assert np.abs(m.encode_tokens(g["ids"], g["mask"]) - g["normed"]).max() < TOL
    m.close()


@pytest.mark.parametrize("tokens", [(1, 5), (1, 32), (2, 20), (3, 33), (8, 16), (5, 40)])
def test_small_batches_take_the_skinny_gemm_path(ctx, oracle, tokens):
    # T = B*L <= 128 tokens runs the K-split skinny GEMM (1, 2 or 4 row tiles); T = 200 the tiled one
    B, L = tokens
    desc = dict(vocab=400, hidden=256, layers=2, heads=8, inter=512, max_pos=64, eps=1e-12, pooling=0, normalize=1)
    m = make_model(ctx, desc, seed=9)
    rng = np.random.default_rng(B * 100 + L)
    ids = rng.integers(1, 400, (B, L)).astype(np.int64)
    mask = np.ones_like(ids)
    mask[-1, max(1, L // 2):] = 0
    ids *= mask
    out = m.encode_tokens(ids, mask)
    oout, _ = oracle.encode_tokens(desc, m.state_dict(), ids, mask)
    assert np.abs(out - oout).max() < 2e-5
    m.close()


def test_graph_replay_matches_eager(ctx, oracle):
    # small forwards are captured into a hipGraph on their second occurrence and replayed afterwards:
    # same bits as the eager launches, also after the inputs (and the batch shape) change
    desc = dict(vocab=400, hidden=256, layers=2, heads=8, inter=512, max_pos=64, eps=1e-12, pooling=0, normalize=1)
    m = make_model(ctx, desc, seed=21)
    sd = m.state_dict()
    rng = np.random.default_rng(0)
    for shape in [(1, 12), (2, 7), (1, 12)]:
        outs = []
        for rep in range(4):  # eager, capture+launch, replay, replay
            ids = rng.integers(1, 400, shape).astype(np.int64)
            mask = np.ones_like(ids)
            out = m.encode_tokens(ids, mask)
            oout, ohid = oracle.encode_tokens(desc, sd, ids, mask, want_hidden=True)
            assert np.abs(out - oout).max() < 2e-5, (shape, rep)
            np.testing.assert_allclose(m.debug_hidden(2, *shape), ohid[-1], atol=1e-4)
            outs.append((ids, out))
        again = m.encode_tokens(outs[0][0], np.ones_like(outs[0][0]))
        np.testing.assert_array_equal(again, outs[0][1])  # replay == eager, bit for bit
    # growing the workspace drops the graphs; the small shape still works afterwards
    big = rng.integers(1, 400, (40, 60)).astype(np.int64)
    m.encode_tokens(big, np.ones_like(big))
    ids = rng.integers(1, 400, (1, 12)).astype(np.int64)
    for _ in range(3):
        out = m.encode_tokens(ids, np.ones_like(ids))
    oout, _ = oracle.encode_tokens(desc, sd, ids, np.ones_like(ids))
    assert np.abs(out - oout).max() < 2e-5
    m.close()


def test_workspace_reuse_across_shapes(ctx, oracle):
    # one Model, shapes that share a padded token count but not a token count (B=4,L=33 then B=4,L=60:
    # both pad to 4*64), then a smaller and a larger batch: every buffer must fit every call
    desc = dict(vocab=400, hidden=128, layers=2, heads=4, inter=256, max_pos=128, eps=1e-12, pooling=0, normalize=1)
    m = make_model(ctx, desc, seed=11)
    sd = m.state_dict()
    rng = np.random.default_rng(5)
    for B, L in [(4, 33), (4, 60), (1, 64), (4, 33), (9, 100), (2, 7)]:
        toks = [list(rng.integers(1, 400, L if i == 0 else int(rng.integers(1, L + 1)))) for i in range(B)]
        ids, mask = m.generate_token_tensors(toks)
        assert ids.shape == (B, L)
        out = m.encode_tokens(ids, mask)
        oout, ohid = oracle.encode_tokens(desc, sd, ids, mask, want_hidden=True)
        np.testing.assert_allclose(out, oout, atol=2e-5, err_msg=f"B={B} L={L}")
        np.testing.assert_allclose(m.debug_hidden(2, B, L), ohid[-1], atol=1e-4)
    m.close()


def test_token_ids_outside_the_vocabulary_are_refused(ctx):
    m = make_model(ctx, dict(vocab=300, hidden=128, layers=1, heads=4, inter=256, max_pos=64), seed=1)
    ids = np.array([[5, 299, 7]], np.int64)
    m.encode_tokens(ids, np.ones_like(ids))
    for bad in (300, -1, 10**9):
        ids2 = ids.copy()
        ids2[0, 1] = bad
        with pytest.raises(pa.ModelError) as e:
            m.encode_tokens(ids2, np.ones_like(ids2))
        assert "vocabulary" in str(e.value)
    m.close()
