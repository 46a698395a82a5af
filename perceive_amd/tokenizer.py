"""Host-side mirror of the tokenizer calls of perceive-core (model/tokenize.rs:60-77,
model/highlight.rs:32-38) over the C ABI's WordPiece tokenizer (csrc/tokenizer.cpp).
rust_tokenizers names are kept: `encode_list` returns objects with `token_ids`, `token_offsets`
(char offsets or None for special tokens) and `special_tokens_mask`."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import _ffi


@dataclass
class TokenizedInput:
    token_ids: List[int]
    token_offsets: List[Optional[Tuple[int, int]]]
    special_tokens_mask: List[int]


class BertTokenizer:
    """TokenizerOption::from_file(Bert, vocab, lower_case, strip_accents) — model.rs:96-113."""

    def __init__(self, vocab_path, lower_case=True, strip_accents=None):
        self._h = C.c_void_p()
        sa = -1 if strip_accents is None else (1 if strip_accents else 0)
        _ffi.check(_ffi.lib().pcv_tokenizer_create(str(vocab_path).encode(), 1 if lower_case else 0, sa, C.byref(self._h)))
        ids = [C.c_int64() for _ in range(4)]
        _ffi.check(_ffi.lib().pcv_tokenizer_special_ids(self._h, *[C.byref(x) for x in ids]))
        self.pad_id, self.unk_id, self.cls_id, self.sep_id = [x.value for x in ids]

    @classmethod
    def _borrowed(cls, handle):
        """A view of a tokenizer some other handle owns (a Model built from a directory): never destroyed here."""
        t = cls.__new__(cls)
        t._h = C.c_void_p(handle)
        t._borrowed_handle = True
        ids = [C.c_int64() for _ in range(4)]
        _ffi.check(_ffi.lib().pcv_tokenizer_special_ids(t._h, *[C.byref(x) for x in ids]))
        t.pad_id, t.unk_id, t.cls_id, t.sep_id = [x.value for x in ids]
        return t

    def get_pad_id(self):
        """tokenize.rs:19: `self.tokenizer.get_pad_id().unwrap_or(0)` at the call site."""
        return self.pad_id if self.pad_id >= 0 else None

    @property
    def vocab_size(self):
        n = C.c_int()
        _ffi.check(_ffi.lib().pcv_tokenizer_vocab_size(self._h, C.byref(n)))
        return n.value

    def encode(self, text, max_len):
        b = text.encode("utf-8")
        cap = min(max_len, len(b) + 2)  # a token covers at least one byte
        while True:
            ids = np.empty(cap, np.int64)
            beg = np.empty(cap, np.int32)
            end = np.empty(cap, np.int32)
            sp = np.empty(cap, np.uint8)
            n = C.c_int()
            st = _ffi.lib().pcv_tokenizer_encode(
                self._h, b, len(b), int(max_len), _ffi.i64p(ids), _ffi.i32p(beg),
                _ffi.i32p(end), _ffi.u8p(sp), cap, C.byref(n))
            if st != 0 and n.value > cap:
                cap = n.value
                continue
            _ffi.check(st)
            k = n.value
            offs = [None if sp[i] else (int(beg[i]), int(end[i])) for i in range(k)]
            return TokenizedInput([int(x) for x in ids[:k]], offs, [int(x) for x in sp[:k]])

    def encode_batch_ids(self, inputs, max_len, pad_id=0, n_threads=0):
        """All of `inputs` at once, in C++ threads: (ids [n, max_len] int64 right-padded with pad_id,
        lens [n] int32).  What Model.tokenize needs (ids only); same tokens as encode_list."""
        texts = [t.encode("utf-8") for t in inputs]
        n = len(texts)
        ids = np.empty((n, int(max_len)), np.int64)
        lens = np.zeros(n, np.int32)
        if n:
            arr = (C.c_char_p * n)(*texts)
            nb = (C.c_size_t * n)(*[len(b) for b in texts])
            _ffi.check(_ffi.lib().pcv_tokenizer_encode_batch(self._h, arr, nb, n, int(max_len), int(pad_id), _ffi.i64p(ids),
                                                             _ffi.i32p(lens), int(n_threads)))
        return ids, lens

    def encode_list(self, inputs, max_len, truncation_strategy="LongestFirst", stride=0):
        """rust_tokenizers `encode_list` as called at tokenize.rs:64-75 / highlight.rs:32-38."""
        if truncation_strategy != "LongestFirst" or stride != 0:
            raise ValueError("only TruncationStrategy::LongestFirst with stride 0 is used by the reference")
        return [self.encode(t, max_len) for t in inputs]

    def close(self):
        if self._h:
            if not getattr(self, "_borrowed_handle", False):
                _ffi.lib().pcv_tokenizer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RobertaTokenizer(BertTokenizer):
    """TokenizerOption::from_file(Roberta, vocab.json, merges.txt, .., add_prefix_space) — the byte-level BPE
    tokenizer of the RoBERTa-family entries of the reference's model list (all-distilroberta-v1).  Same methods
    as BertTokenizer (`encode`, `encode_list`, `encode_batch_ids`); cls / sep / pad are <s> / </s> / <pad>."""

    def __init__(self, vocab_json_path, merges_path, add_prefix_space=False):
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().pcv_tokenizer_create_bpe(str(vocab_json_path).encode(), str(merges_path).encode(),
                                                       1 if add_prefix_space else 0, C.byref(self._h)))
        ids = [C.c_int64() for _ in range(4)]
        _ffi.check(_ffi.lib().pcv_tokenizer_special_ids(self._h, *[C.byref(x) for x in ids]))
        self.pad_id, self.unk_id, self.cls_id, self.sep_id = [x.value for x in ids]


class AlbertTokenizer(BertTokenizer):
    """TokenizerOption::from_file(Albert, spiece.model, lower_case, strip_accents) — the SentencePiece unigram
    tokenizer of the ParaphraseAlbertSmallV2 entry of the reference's model list (configs.rs:35).  Same methods as
    BertTokenizer; cls / sep / pad are [CLS] / [SEP] / <pad>."""

    def __init__(self, model_path, lower_case=True, strip_accents=None):
        self._h = C.c_void_p()
        sa = -1 if strip_accents is None else (1 if strip_accents else 0)
        _ffi.check(_ffi.lib().pcv_tokenizer_create_sentencepiece(str(model_path).encode(), 1 if lower_case else 0, sa, C.byref(self._h)))
        ids = [C.c_int64() for _ in range(4)]
        _ffi.check(_ffi.lib().pcv_tokenizer_special_ids(self._h, *[C.byref(x) for x in ids]))
        self.pad_id, self.unk_id, self.cls_id, self.sep_id = [x.value for x in ids]


def nfkc(text):
    """Unicode NFKC as the library computes it (pcv_unicode_nfkc)."""
    b = text.encode("utf-8")
    n = C.c_size_t()
    _ffi.check(_ffi.lib().pcv_unicode_nfkc(b, len(b), None, 0, C.byref(n)))
    out = C.create_string_buffer(max(1, n.value))
    _ffi.check(_ffi.lib().pcv_unicode_nfkc(b, len(b), out, n.value, C.byref(n)))
    return out.raw[: n.value].decode("utf-8")
