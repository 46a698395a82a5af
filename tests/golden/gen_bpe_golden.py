#!/usr/bin/env python3
"""Golden vectors for the byte-level BPE tokenizer (RoBERTa-family checkpoints of the reference's model list,
e.g. all-distilroberta-v1).  No real vocabulary exists offline, so a small one is TRAINED here with the Hugging
Face `tokenizers` library (deterministic for a fixed corpus) and the same library produces the expected ids and
character offsets:

    python tests/golden/gen_bpe_golden.py   ->  bpe_vocab.json, bpe_merges.txt, bpe_golden.json
"""
import json
import os

from tokenizers import ByteLevelBPETokenizer

HERE = os.path.dirname(os.path.abspath(__file__))

CORPUS = [
    "Hello world, this is a test of the byte level BPE tokenizer.",
    "The quick brown fox jumps over the lazy dog 1234 times!",
    "Ünïcödé têxt with ümlauts, naïve café, and 中文 characters and emoji 🙂.",
    "don't can't won't it's we're they've I'll he'd — she'll 'tis",
    "Searching embeddings: cosine similarity, dot product, nearest neighbours.",
    "numbers 3.14159 2,718 1e-9 0x7f 100% $42.00 #hashtag @mention",
    "tabs\tand\nnewlines\r\nand   multiple   spaces",
    "Русский текст и ελληνικά γράμματα ١٢٣ ४५६",
] * 40

CASES = [
    "", " ", "  ", "a", " a", "a ", "Hello world", " Hello world", "Hello world ", "Hello  world", "Hello   world  ",
    "don't stop", "DON'T STOP", "it's 'quoted' isn't it?", "they've we're I'll he'd she's I'm", "'s't're've'm'll'd",
    "123 4567 89", "abc123def", "3.14 and 2,718", "100%!!! ???", "a-b_c+d=e", "e-mail: foo.bar@example.com",
    "tab\there", "line\nbreak", "crlf\r\nend", "trailing newline\n", "\n\nleading newlines", "a \n b", "a\n\n\nb",
    "space before newline \nnext", "nbsp here", "ideographic　space", "thin space", "zero​width",
    "Ünïcödé", "naïve café", "中文字符", "日本語のテキスト", "한국어 텍스트", "emoji 🙂 here", "🙂🙂", "mixed中文and🙂English",
    "Русский текст", "ελληνικά", "١٢٣ ४५६ ⅓ Ⅻ ²", "combining é ä", "ǅ titlecase ʰ modifier ª ordinal",
    "The quick brown fox jumps over the lazy dog.", "word " * 80, "x" * 300, "!?" * 40,
    "<s> literal specials </s> <pad> <unk> <mask>", "  leading and trailing  ", "\t\ttabs\t\t", " \n \t mixed whitespace \n ",
]


def main():
    t = ByteLevelBPETokenizer()
    t.train_from_iterator(CORPUS, vocab_size=700, min_frequency=1,
                          special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>"])
    t.save_model(HERE, "bpe")  # bpe-vocab.json / bpe-merges.txt
    os.replace(os.path.join(HERE, "bpe-vocab.json"), os.path.join(HERE, "bpe_vocab.json"))
    os.replace(os.path.join(HERE, "bpe-merges.txt"), os.path.join(HERE, "bpe_merges.txt"))
    out = []
    for prefix in (False, True):
        tok = ByteLevelBPETokenizer(os.path.join(HERE, "bpe_vocab.json"), os.path.join(HERE, "bpe_merges.txt"),
                                    add_prefix_space=prefix)
        for text in CASES:
            e = tok.encode(text)
            out.append({"text": text, "add_prefix_space": prefix, "ids": e.ids, "offsets": [list(o) for o in e.offsets]})
    with open(os.path.join(HERE, "bpe_golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)
    print(len(out), "cases;", sum(len(c["ids"]) for c in out), "tokens")


if __name__ == "__main__":
    main()
