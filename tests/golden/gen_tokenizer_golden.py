"""Generates tests/golden/tokenizer_vocab.txt + tokenizer_golden.json with Hugging Face's
BertTokenizer (the Python port of Google BERT's tokenization.py, the same algorithm
rust_tokenizers' BertTokenizer ports — tokenize.rs:64-75) for ids, and BertTokenizerFast for the
char offsets (TokenIdsWithOffsets::token_offsets in rust_tokenizers).  The vocab is synthetic (no
real vocab.txt exists offline).  Run once, offline:  python tests/golden/gen_tokenizer_golden.py
"""
import json
import os

os.environ["HF_HUB_OFFLINE"] = "1"
from transformers import BertTokenizer, BertTokenizerFast

HERE = os.path.dirname(os.path.abspath(__file__))


def build_vocab():
    v = ["[PAD]"] + [f"[unused{i}]" for i in range(5)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789") + list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") + list("—…«»¿¡。、！？")
    v += chars + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    words = ("the of and to in a is that for it as was with be by on not he this are or his from at which but have an "
             "they you were her all she there would their we him been has when who will more no if out so said what up "
             "its about into than them can only other new some could time these two may then do first any my now such "
             "like our over man me even most made after also did many before must through years where much your way "
             "well down should because each just those people how too little state good very make world still own see "
             "men work long get here between both life being under never day same another know while last might us "
             "great old year off come since against go came right used take three search vector embedding cosine "
             "similarity query document model sentence hello cafe naive angstrom strasse viet nam istanbul tokyo "
             "don stop really").split()
    pieces = "##s ##ing ##ed ##ly ##er ##est ##tion ##ment ##ness ##able ##al ##ic ##ize ##ous ##ful ##less ##t ##n ##e".split()
    other = ["привет", "мир", "при", "##вет", "αθηνα", "αθ", "##ηνα", "東", "京", "世", "界", "タ", "ワ", "ー", "##ー",
             "한", "국", "어", "ᄒ", "ß", "strasse", "stra", "##ße", "ı", "i", "ﬁ", "abc", "æ", "ø", "##ø", "łodz", "ł", "##odz",
             "3", "14", "000", "1", "##00", "e", "##mail", "e-mail"]
    seen, out = set(), []
    for t in v + words + pieces + other:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


TEXTS = [
    "Hello world",
    "The QUICK search of embeddings, really?",
    "don't—stop... (really?) «maybe» ¿qué? ¡sí!",
    "Café naïve Ångström Żółć façade",
    "東京タワーhello世界",
    "Привет МИР",
    "ΑΘΗΝΑ αθηνα",
    "a\x00b​c�d \t tabs\nand\r\nnewlines nbsp　ideographic",
    "x" * 120 + " after a very long word",
    "unknown 🙂 emoji and ☃ snowman",
    "text with [SEP] and [MASK] inside [CLS]",
    "İstanbul ISTANBUL ıspanak",
    "Straße STRASSE",
    "Việt Nam",
    "한국어 korean",
    "ＡＢＣ！ fullwidth",
    "ﬁne ligature ǅ titlecase",
    "3.14 and 1,000 e-mail",
    "",
    "   ",
    "searching documents embeddingly nonmatchingzzzq",
    "Łódź łodz",
    "the " * 40,
]
MAX_LENS = [256, 16, 5, 2]


def main():
    vocab = build_vocab()
    vp = os.path.join(HERE, "tokenizer_vocab.txt")
    with open(vp, "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    cases = []
    for lower, strip in [(True, None), (False, None), (True, False), (False, True)]:
        slow = BertTokenizer(vp, do_lower_case=lower, strip_accents=strip)
        fast = BertTokenizerFast(vp, do_lower_case=lower, strip_accents=strip)
        for text in TEXTS:
            for ml in MAX_LENS:
                ids = slow.encode(text, add_special_tokens=True, max_length=ml, truncation=True)
                enc = fast(text, add_special_tokens=True, max_length=ml, truncation=True, return_offsets_mapping=True,
                           return_special_tokens_mask=True)
                cases.append({
                    "lower": lower, "strip": strip, "text": text, "max_len": ml, "ids": ids,
                    "fast_ids": enc["input_ids"],
                    "offsets": [None if s else list(o) for o, s in zip(enc["offset_mapping"], enc["special_tokens_mask"])],
                    "special": enc["special_tokens_mask"],
                })
    with open(os.path.join(HERE, "tokenizer_golden.json"), "w", encoding="utf-8") as f:
        json.dump({"cases": cases}, f, ensure_ascii=True)
    nd = sum(1 for c in cases if c["ids"] != c["fast_ids"])
    print(len(vocab), "vocab entries;", len(cases), "cases;", nd, "where HF slow and fast disagree")


if __name__ == "__main__":
    main()
