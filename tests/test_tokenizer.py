"""CPU: the C++ WordPiece tokenizer (csrc/tokenizer.cpp, host code like the reference's tokenizer)
against Hugging Face BertTokenizer / BertTokenizerFast vectors (tests/golden/gen_tokenizer_golden.py).
Bit-exact: ids, special-token masks and char offsets."""
import json
import os

import numpy as np
import pytest

import perceive_amd as pa


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "tokenizer_golden.json"), encoding="utf-8") as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="module")
def vocab_path(golden_dir):
    return os.path.join(golden_dir, "tokenizer_vocab.txt")


def test_ids_offsets_and_masks_match_hf(golden, vocab_path):
    toks = {}
    bad = []
    for c in golden:
        key = (c["lower"], c["strip"])
        if key not in toks:
            toks[key] = pa.BertTokenizer(vocab_path, lower_case=c["lower"], strip_accents=c["strip"])
        enc = toks[key].encode(c["text"], c["max_len"])
        exp_off = [None if o is None else tuple(o) for o in c["offsets"]]
        if enc.token_ids != c["ids"] or enc.special_tokens_mask != c["special"] or enc.token_offsets != exp_off:
            bad.append((key, c["text"][:40], c["max_len"], enc.token_ids, c["ids"], enc.token_offsets, exp_off))
    assert not bad, f"{len(bad)} of {len(golden)} cases differ; first: {bad[0]}"
    assert len(golden) >= 300


def test_batch_entry_point_matches_hf_ids(golden, vocab_path):
    # pcv_tokenizer_encode_batch (threaded, what Model.tokenize uses): same ids as the golden HF cases for every
    # max_len, rows right-padded with the pad id, any thread count; empty input list
    by_cfg = {}
    for c in golden:
        by_cfg.setdefault((c["lower"], c["strip"], c["max_len"]), []).append(c)
    checked = 0
    for (lower, strip, max_len), cases in by_cfg.items():
        tok = pa.BertTokenizer(vocab_path, lower_case=lower, strip_accents=strip)
        texts = [c["text"] for c in cases]
        for nt in (1, 3, 0):
            ids, lens = tok.encode_batch_ids(texts, max_len, pad_id=7, n_threads=nt)
            assert ids.shape == (len(cases), max_len)
            for i, c in enumerate(cases):
                assert lens[i] == len(c["ids"]) and list(ids[i, : lens[i]]) == c["ids"], (c["text"][:40], nt)
                assert (ids[i, lens[i]:] == 7).all()
        checked += len(cases)
        ids0, lens0 = tok.encode_batch_ids([], max_len)
        assert ids0.shape == (0, max_len) and lens0.shape == (0,)
    assert checked == len(golden)
    with pytest.raises(pa.PcvError):
        tok.encode_batch_ids(["a"], 1)  # no room for [CLS] [SEP]


def test_encode_list_and_padding_layout(vocab_path):
    # Model::tokenize (tokenize.rs:60-77): encode_list -> token_ids -> generate_token_tensors
    t = pa.BertTokenizer(vocab_path)
    enc = t.encode_list(["hello world", "the search of embeddings, really?"], 256)
    assert [e.token_ids[0] for e in enc] == [t.cls_id] * 2 and [e.token_ids[-1] for e in enc] == [t.sep_id] * 2
    assert t.get_pad_id() == 0 and t.vocab_size == 340
    L = max(len(e.token_ids) for e in enc)
    ids = np.full((2, L), t.pad_id, np.int64)
    for i, e in enumerate(enc):
        ids[i, : len(e.token_ids)] = e.token_ids
    mask = (ids != t.pad_id).astype(np.int64)  # tokenize.rs:36-46
    assert mask.sum(1).tolist() == [len(e.token_ids) for e in enc]
    with pytest.raises(ValueError):
        t.encode_list(["x"], 8, truncation_strategy="OnlyFirst")


def test_errors(tmp_path, vocab_path):
    with pytest.raises(pa.PcvError) as e:
        pa.BertTokenizer(str(tmp_path / "missing.txt"))
    assert e.value.status == 4
    p = tmp_path / "v.txt"
    p.write_text("a\nb\n")
    with pytest.raises(pa.PcvError):
        pa.BertTokenizer(str(p))  # no [UNK]/[CLS]/[SEP]
    t = pa.BertTokenizer(vocab_path)
    with pytest.raises(pa.PcvError):
        t.encode("hello", 1)


def test_byte_level_bpe_matches_the_tokenizers_library(golden_dir):
    # RoBERTa-family tokenizer: ids and character offsets against Hugging Face `tokenizers` on a vocabulary
    # trained by tests/golden/gen_bpe_golden.py (110 cases: contractions, digits, whitespace runs of every kind,
    # NBSP / ideographic space, CJK, emoji split across tokens, both add_prefix_space settings)
    cases = json.load(open(os.path.join(golden_dir, "bpe_golden.json"), encoding="utf-8"))
    toks = {p: pa.RobertaTokenizer(os.path.join(golden_dir, "bpe_vocab.json"), os.path.join(golden_dir, "bpe_merges.txt"),
                                   add_prefix_space=p) for p in (False, True)}
    assert toks[False].cls_id == 0 and toks[False].pad_id == 1 and toks[False].sep_id == 2 and toks[False].unk_id == 3
    bad = []
    for c in cases:
        enc = toks[c["add_prefix_space"]].encode(c["text"], 4096)
        ids = enc.token_ids[1:-1]
        offs = [list(o) for o in enc.token_offsets[1:-1]]
        if ids != c["ids"] or offs != c["offsets"] or enc.token_ids[0] != 0 or enc.token_ids[-1] != 2:
            bad.append((c["text"][:40], c["add_prefix_space"], ids[:12], c["ids"][:12], offs[:6], c["offsets"][:6]))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ; first: {bad[0]}"
    assert len(cases) >= 100
    # truncation keeps <s> ... </s>; the batch entry point pads with <pad>
    e = toks[False].encode("word " * 80, 16)
    assert len(e.token_ids) == 16 and e.token_ids[0] == 0 and e.token_ids[-1] == 2
    ids, lens = toks[False].encode_batch_ids(["Hello world", "", "x" * 300], 12, pad_id=1, n_threads=2)
    assert list(lens) == [4, 2, 12] and list(ids[1]) == [0, 2] + [1] * 10 and ids[0][3] == 2


def test_nfkc_matches_unicodedata():
    """The NFKC the SentencePiece path applies (rust_tokenizers' decompose_nfkc) against Python's unicodedata, which
    the tables were generated from: every code point that has a compatibility mapping or a combining class, and
    strings that exercise canonical reordering and composition."""
    import random
    import unicodedata as ud

    singles = [chr(cp) for cp in range(0xA0, 0x30000) if not 0xD800 <= cp <= 0xDFFF
               and (ud.normalize("NFKC", chr(cp)) != chr(cp) or ud.combining(chr(cp)))]
    assert len(singles) > 5000
    joined = "\n".join(singles)  # '\n' is a starter: one call checks them all independently
    assert pa.nfkc(joined) == ud.normalize("NFKC", joined)
    rnd = random.Random(1)
    marks = [chr(c) for c in range(0x300, 0x370)] + [chr(0x5B0 + i) for i in range(20)] + list("़゙゚ཱིུᅡᆨ")
    bases = list("aeiouAEIOUnNcCsSyYとはか가한ㄱᄀ") + ["ﬁ", "㌀", "①", "ǆ", "ｶ", "ﾞ", "Å", "Ω", "ೆ", "ෙ"]
    for _ in range(3000):
        s = "".join(rnd.choice(bases + marks * 2) for _ in range(rnd.randint(1, 8)))
        assert pa.nfkc(s) == ud.normalize("NFKC", s), [hex(ord(c)) for c in s]
    assert pa.nfkc("") == "" and pa.nfkc("plain ascii") == "plain ascii"


def test_sentencepiece_matches_the_sentencepiece_library(golden_dir):
    """AlbertTokenizer (SentencePiece unigram) against ids the sentencepiece library's own Viterbi produced for the
    same prepared text + ALBERT's digit rule (tests/golden/gen_spm_golden.py), in all three normalisation modes."""
    g = json.load(open(os.path.join(golden_dir, "spm_golden.json"), encoding="utf-8"))
    model = os.path.join(golden_dir, "spiece.model")
    toks = {}
    assert len(g["cases"]) > 100
    for c in g["cases"]:
        key = (c["lower"], c["strip"])
        if key not in toks:
            toks[key] = pa.AlbertTokenizer(model, lower_case=c["lower"], strip_accents=c["strip"])
        enc = toks[key].encode(c["text"], c["max_len"])
        assert enc.token_ids == c["ids"], (c["text"], key, c["max_len"])
        assert enc.special_tokens_mask == [1] + [0] * (len(c["ids"]) - 2) + [1]
    t = toks[(True, True)]
    assert (t.pad_id, t.unk_id, t.cls_id, t.sep_id) == (0, 1, 2, 3) and t.vocab_size == g["vocab_size"]
    # the batch entry point writes the same ids, padded with <pad>
    texts = sorted({c["text"] for c in g["cases"]})
    ids, lens = t.encode_batch_ids(texts, 24, pad_id=t.pad_id, n_threads=3)
    for row, n, text in zip(ids, lens, texts):
        assert list(row[:n]) == t.encode(text, 24).token_ids and (row[n:] == 0).all()


def test_sentencepiece_offsets_unknowns_and_errors(golden_dir, tmp_path):
    t = pa.AlbertTokenizer(os.path.join(golden_dir, "spiece.model"))
    # offsets are character positions in the caller's text; the inserted leading U+2581 covers nothing
    text = "Hello  wörld, in 1999, ok"
    enc = t.encode(text, 64)
    spans = [o for o in enc.token_offsets if o is not None]
    assert spans[0][0] == 0 and spans[-1][1] == len(text)
    assert all(a <= b for a, b in spans) and all(spans[i][1] <= spans[i + 1][0] or spans[i][0] <= spans[i + 1][0] for i in range(len(spans) - 1))
    covered = "".join(text[a:b] for a, b in spans)
    assert covered.replace(" ", "") == text.replace(" ", "")  # nothing but whitespace falls between pieces
    # a character outside the model: one <unk> per character, the pieces around it unaffected (rust_tokenizers'
    # decode_forward restarts the score behind it; the sentencepiece library would merge a run into one <unk>)
    a = t.encode("hello world", 64).token_ids
    b = t.encode("hello 東京 world", 64).token_ids
    assert b.count(t.unk_id) == 2 and [x for x in b if x != t.unk_id][:2] == a[:2] and b[-2:] == a[-2:]
    with pytest.raises(pa.PcvError):
        pa.AlbertTokenizer(str(tmp_path / "missing.model"))
    (tmp_path / "junk.model").write_bytes(b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff")
    with pytest.raises(pa.PcvError):
        pa.AlbertTokenizer(str(tmp_path / "junk.model"))
    (tmp_path / "vocab.txt").write_text("[PAD]\n[UNK]\n")
    with pytest.raises(pa.PcvError):  # a text vocabulary is not a SentencePiece model
        pa.AlbertTokenizer(str(tmp_path / "vocab.txt"))
