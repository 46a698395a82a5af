"""Generates tests/golden/spiece.model + spm_golden.json for the SentencePiece (ALBERT) tokenizer.

No `spiece.model` of paraphrase-albert-small-v2 exists offline, so a small unigram model is trained here with the
sentencepiece library itself, with ALBERT's special-piece layout (<pad> 0, <unk> 1, [CLS] 2, [SEP] 3, [MASK] 4).
Expected ids come from the library's own Viterbi (`SentencePieceProcessor.encode`) applied to the text as the
reference's tokenizer prepares it — rust_tokenizers' AlbertTokenizer: clean_text, NFKC, lower-case, strip accents,
whitespace -> U+2581, a leading U+2581 — followed by ALBERT's "<digit>," re-split (tokenization_albert.py of the
original ALBERT release).  The model is trained with the identity normaliser so that the library adds nothing of
its own; the normalisation steps are restated below with Python's unicodedata.
Texts stay inside the model's alphabet: rust_tokenizers and the library treat unknown characters differently
(one <unk> per character there, merged runs here); tests/test_tokenizer.py pins that case by hand.
Run once:   python tests/golden/gen_spm_golden.py
"""
import io
import json
import os
import pydoc
import unicodedata as ud

import sentencepiece as spm

HERE = os.path.dirname(os.path.abspath(__file__))
U = "▁"


def is_control(c):
    return c not in "\t\n\r" and ud.category(c) in ("Cc", "Cf")


def is_ws(c):
    return c in " \t\n\r" or ud.category(c) == "Zs"


def prepare(text, lower, strip):
    """rust_tokenizers AlbertTokenizer::tokenize_to_tokens up to the segmentation."""
    t = "".join(" " if is_ws(c) else c for c in text if not (c == "\0" or c == "�" or is_control(c)))
    t = ud.normalize("NFKC", t)
    if lower:
        t = "".join(c.lower() for c in t)  # per character, like char::to_lowercase
    if strip:
        t = "".join(c for c in ud.normalize("NFD", t) if ud.category(c) != "Mn")
    t = "".join(U if is_ws(c) else c for c in t)
    if not t.startswith(U):
        t = U + t
    return t


def segment(sp, prepared):
    # the library prepends the dummy prefix itself and escapes ' ' to U+2581: hand it the text that makes its
    # internal string equal to `prepared`
    assert prepared.startswith(U)
    if prepared == U:  # empty input: rust_tokenizers still segments its inserted U+2581 (the library returns nothing for "")
        return [U]
    return sp.encode(prepared[1:].replace(U, " "), out_type=str)


def albert_pieces(sp, prepared):
    out = []
    for piece in segment(sp, prepared):
        if len(piece) > 1 and piece[-1] == "," and piece[-2].isdigit():
            cur = sp.encode(piece[:-1].replace(U, ""), out_type=str)
            if piece[0] != U and cur[0][0] == U:
                if len(cur[0]) == 1:
                    cur = cur[1:]
                else:
                    cur[0] = cur[0][1:]
            cur.append(piece[-1])
            out.extend(cur)
        else:
            out.append(piece)
    return out


def corpus():
    import collections, json as js, os as o, re, string, textwrap, itertools, functools, heapq

    lines = []
    for mod in (collections, js, o, re, string, textwrap, itertools, functools, heapq):
        text = pydoc.render_doc(mod, renderer=pydoc.plaintext)
        for ln in text.splitlines():
            ln = " ".join(ln.split())
            if len(ln) > 20:
                lines.append(prepare(ln, True, True)[1:].replace(U, " "))
    extra = ["in 1999, 2000, and 2021, about 1,000 or 10,000 items", "hello world", "the search of embeddings, really?",
             "cafe naive angstrom facade", "e-mail 3.14 and 42, 7, 128, 256, 1024,", "abc fine ligature", "strasse istanbul"]
    return lines + extra * 30


TEXTS = [
    "Hello world",
    "The QUICK search of embeddings, really?",
    "Café naïve Ångström façade",
    "in 1999, about 1,000 items; then 2000, 42, and 7,",
    "x2000, y7, z",
    "ＡＢＣ fullwidth ﬁne ligature",
    "a\x00b​c d \t tabs\nand\r\nnewlines nbsp　ideographic",
    "double  spaces   here ",
    "  leading and trailing  ",
    "İstanbul ISTANBUL",
    "",
    "   ",
    "returns a new string object with the characters reversed and joined",
    "the " * 40,
    "supercalifragilisticexpialidocious nonmatchingzzzq",
]
MAX_LENS = [256, 16, 5, 2]


def main():
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(corpus()), model_writer=model, vocab_size=700, model_type="unigram",
                                   normalization_rule_name="identity", remove_extra_whitespaces=False, add_dummy_prefix=True,
                                   pad_id=0, unk_id=1, bos_id=-1, eos_id=-1, pad_piece="<pad>", unk_piece="<unk>",
                                   control_symbols=["[CLS]", "[SEP]", "[MASK]"], character_coverage=1.0, hard_vocab_limit=False,
                                   split_digits=False, minloglevel=2)
    blob = model.getvalue()
    open(os.path.join(HERE, "spiece.model"), "wb").write(blob)
    sp = spm.SentencePieceProcessor(model_proto=blob)
    assert [sp.piece_to_id(p) for p in ("<pad>", "<unk>", "[CLS]", "[SEP]", "[MASK]")] == [0, 1, 2, 3, 4]
    alphabet = set("".join(sp.id_to_piece(i) for i in range(5, sp.get_piece_size())))
    cases = []
    for lower, strip in [(True, True), (True, False), (False, False)]:
        for text in TEXTS:
            prepared = prepare(text, lower, strip)
            if not set(prepared) <= alphabet:
                if lower and strip:
                    raise SystemExit(f"text outside the model's alphabet: {text!r} {sorted(set(prepared) - alphabet)}")
                continue  # upper-case letters / accents the model never saw: only checked in the modes that remove them
            pieces = albert_pieces(sp, prepared)
            ids = [sp.piece_to_id(p) for p in pieces]
            for ml in MAX_LENS:
                cases.append({"text": text, "lower": lower, "strip": strip, "max_len": ml, "pieces": pieces[: ml - 2],
                              "ids": [2] + ids[: ml - 2] + [3]})
    json.dump({"sentencepiece": spm.__version__, "vocab_size": sp.get_piece_size(), "cases": cases},
              open(os.path.join(HERE, "spm_golden.json"), "w"), ensure_ascii=False, indent=0)
    n_comma = sum(1 for c in cases if "," in c["pieces"] and c["max_len"] == 256)
    print("pieces", sp.get_piece_size(), "cases", len(cases), "comma cases", n_comma, "model bytes", len(blob))
    for c in cases[:8:4] + [c for c in cases if "1999" in c["text"]][:1]:
        print(c["text"][:40], c["pieces"][:24])


if __name__ == "__main__":
    main()
