"""GPU: text in -> embedding / highlight out (Model::encode, Model::highlight) against the CPU
restatement (tests/oracle_highlight.py: HF tokenizer + C oracle encoder + numpy)."""
import os

import numpy as np
import pytest

import perceive_amd as pa

pytestmark = pytest.mark.gpu

FILL = "the of and to in a is that for it as was with be by on not he this are or his from at which but have an they "
DOCS = [
    FILL * 3 + "people ask how good is the search model in the world today " + FILL * 3,
    "The search of embeddings is made by cosine similarity. A query vector and a document vector are used; "
    "the model will make each sentence into a vector first. Then people search many documents at the same time, "
    "and the great old world will know how good the new model really is, because it can take three years of work.",
    "Hello world. This document is about Tokyo and Istanbul and a cafe where people work each day. "
    "They said the time was right, and so the work could go on for years. Still, no one would know where it came from.",
    "short one",
    "",
    "Hello hello hello hello hello hello hello hello hello hello hello hello hello hello hello hello hello hello",
]


@pytest.fixture(scope="module")
def setup(ctx, golden_dir):
    vocab = os.path.join(golden_dir, "tokenizer_vocab.txt")
    tok = pa.BertTokenizer(vocab)
    desc = dict(vocab=tok.vocab_size, hidden=128, layers=2, heads=4, inter=256, max_pos=512, eps=1e-12, pooling=0,
                normalize=1)
    d = pa.make_desc(desc["vocab"], 128, 2, 4, 256, 512, max_seq_length=64)
    m = pa.Model(ctx, d, synthetic_seed=11, tokenizer=tok)
    yield m, desc, vocab
    m.close()


def test_encode_text_matches_oracle(setup, oracle):
    m, desc, vocab = setup
    texts = ["Hello world", "the search of embeddings, really?", "x " * 200]  # last one truncated to max_seq_length
    out = m.encode(texts)
    ids, mask = m.tokenize(texts)
    assert ids.shape[1] == 64 and mask[2].sum() == 64 and mask[0].sum() == 4
    oout, _ = oracle.encode_tokens(desc, m.state_dict(), ids, mask)
    assert np.abs(out - oout).max() < 1e-4
    q = pa.encode_query(m, "hello world")  # search.rs:262-264
    np.testing.assert_allclose(q, out[0], atol=1e-6)


def test_highlight_matches_oracle(setup, oracle):
    import oracle_highlight

    m, desc, vocab = setup
    got = m.highlight("how good is the search model", DOCS)
    exp, scores = oracle_highlight.highlight(oracle, desc, m.state_dict(), vocab, 64, "how good is the search model", DOCS)
    assert got == exp
    assert got[3] is None and got[4] is None          # too short for a chunk: no highlight (highlight.rs:122-125)
    for g, d in zip(got, DOCS):
        assert g is None or g in d  # a slice of the document ('' when the reference's end char is missing)
    assert any(g for g in got), got
    # env-overridable chunking (highlight.rs:7-18)
    os.environ["CHUNK_SIZE"], os.environ["CHUNK_OVERLAP"] = "8", "2"
    try:
        got8 = m.highlight("tokyo cafe", DOCS)
        exp8, _ = oracle_highlight.highlight(oracle, desc, m.state_dict(), vocab, 64, "tokyo cafe", DOCS, 8, 2)
        assert got8 == exp8 and got8[4] is None
    finally:
        del os.environ["CHUNK_SIZE"], os.environ["CHUNK_OVERLAP"]
    # the same through explicit arguments, and a Unicode document: offsets are chars, the returned range is bytes
    assert m.highlight("tokyo cafe", DOCS, chunk_size=8, chunk_overlap=2) == exp8
    uni = ["Ünïcödé çafé in Tōkyō — how good is the search model — " + "people work each day and years go on. " * 6 + "naïve café",
           "Ünïcödé çafé in Tōkyō — " + "people work each day and the search model is good " * 3 + "naïve café"]
    got_u = m.highlight("how good is the search model", uni)
    exp_u, _ = oracle_highlight.highlight(oracle, desc, m.state_dict(), vocab, 64, "how good is the search model", uni)
    assert got_u == exp_u and all(g in d for g, d in zip(got_u, uni))
    assert any("ï" in g or "ō" in g or "—" in g for g in got_u), got_u  # a range that crosses multi-byte chars
    with pytest.raises(pa.ModelError):
        m.highlight("q", DOCS, chunk_size=4, chunk_overlap=4)


def test_highlight_many_documents_batches(setup, oracle):
    # more chunks than one forward takes (2048): several batches, same answer per document as alone
    m, desc, vocab = setup
    docs = [DOCS[1] + " " + DOCS[2] * (1 + i % 3) for i in range(120)] + [DOCS[0]] * 80
    got = m.highlight("where did the work come from", docs)
    assert len(got) == 200 and all(g for g in got)
    for i in (0, 1, 2, 119, 120, 199):
        assert got[i] == m.highlight("where did the work come from", [docs[i]])[0]


def test_encode_text_large_batch_and_errors(setup, oracle):
    m, desc, vocab = setup
    texts = [f"document number {i} about the search of embeddings" for i in range(1500)]  # two forwards of <= 1024
    out = m.encode(texts)
    assert out.shape == (1500, 128)
    np.testing.assert_allclose(out[1234], m.encode([texts[1234]])[0], atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
    assert m.encode([]).shape == (0, 128)
    bare = pa.Model(m.ctx, pa.make_desc(desc["vocab"], 128, 1, 4, 256, 64), synthetic_seed=1)
    with pytest.raises(pa.ModelError) as e:
        bare.encode(["no tokenizer"])
    assert "tokenizer" in str(e.value)
    bare.close()
