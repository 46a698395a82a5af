"""Host-side mirror of perceive-core's model module (crates/perceive-core/model.rs,
model/{configs,tokenize,worker,highlight}.rs) over the C ABI.

Everything with logic in it lives in the library: `encode_tokens` (worker.rs:78-106, the device forward),
`encode` (tokenize + forward, pcv_model_encode_text), `highlight` (pcv_model_highlight) and
`Model.from_dir` (Model::new_pretrained, pcv_model_create_from_dir).  This module converts arguments.
"""
import ctypes as C
import enum
import os
import struct

import numpy as np

from . import _ffi
from .context import Context


class SentenceEmbeddingsModelType(enum.Enum):
    """configs.rs:30-39; `.model_id` is the database id mapping of configs.rs:72-83."""

    AllMiniLmL6V2 = 0
    AllMiniLmL12V2 = 1
    DistiluseBaseMultilingualCased = 2
    AllDistilrobertaV1 = 3
    ParaphraseAlbertSmallV2 = 4
    MsMarcoDistilbertDotV5 = 5
    MsMarcoDistilbertBaseTasB = 6
    MsMarcoBertBaseDotV5 = 7

    @property
    def model_id(self):
        return self.value


class ModelError(RuntimeError):
    """model.rs:29-42 (`ModelError`): any failure of the device forward surfaces here."""


def _tokenizer_threads():
    """Host threads for batch tokenization: the CPUs this process may use (affinity mask, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 32))


_COMPUTE = {"f32": _ffi.COMPUTE_F32, "bf16x3": _ffi.COMPUTE_BF16X3, "f16x2": _ffi.COMPUTE_F16X2}


def minilm_l6_desc(compute="f32"):
    d = _ffi.ModelDesc()
    _ffi.lib().pcv_model_desc_minilm_l6(C.byref(d))
    d.compute = _COMPUTE[compute]
    return d


def make_desc(vocab_size, hidden, layers, heads, intermediate, max_positions, type_vocab=2, layer_norm_eps=1e-12,
              pooling="mean", normalize=True, dense_out=0, dense_activation="identity", max_seq_length=256,
              compute="f32"):
    pools = {"mean": _ffi.POOL_MEAN, "cls": _ffi.POOL_CLS, "max": _ffi.POOL_MAX, "mean_sqrt_len": _ffi.POOL_MEAN_SQRT_LEN}
    acts = {"identity": _ffi.ACT_IDENTITY, "tanh": _ffi.ACT_TANH}
    return _ffi.ModelDesc(vocab_size, hidden, layers, heads, intermediate, max_positions, type_vocab, layer_norm_eps,
                          pools[pooling], 1 if normalize else 0, dense_out, acts[dense_activation], max_seq_length,
                          _COMPUTE[compute])


def save_weights(path, tensors):
    """Write the library's flat weight format: "PCVW0001", u32 count, then per tensor
    u32 name_len, name, u64 numel, f32 data (little-endian).  `tensors`: name -> array, HF BERT names
    (what rust-bert's `rust_model.ot` holds after convert_model.py, scripts/install_models.sh:23-36)."""
    with open(path, "wb") as f:
        f.write(b"PCVW0001")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, dtype="<f4")
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<Q", a.size))
            f.write(a.tobytes())


class Model:
    """model.rs:56-65.  `model_type` is kept as the pub field it is in the reference."""

    def __init__(self, ctx: Context, desc=None, weights_path=None, synthetic_seed=0,
                 model_type=SentenceEmbeddingsModelType.AllMiniLmL6V2, pad_token_id=0, tokenizer=None):
        self.ctx = ctx
        self.model_type = model_type
        self.desc = desc if desc is not None else minilm_l6_desc()
        self.tokenizer = tokenizer  # perceive_amd.BertTokenizer (model.rs:61-63)
        if tokenizer is not None and tokenizer.get_pad_id() is not None:
            pad_token_id = tokenizer.get_pad_id()  # tokenize.rs:19
        self.pad_token_id = pad_token_id
        self._h = C.c_void_p()
        wp = weights_path.encode() if weights_path else None
        _ffi.check(_ffi.lib().pcv_model_create(ctx.handle, C.byref(self.desc), wp, int(synthetic_seed), C.byref(self._h)))
        ctx._register(self)
        if tokenizer is not None:  # Model::tokenizer (model.rs:61): the library tokenizes for encode / highlight
            _ffi.check(_ffi.lib().pcv_model_set_tokenizer(self._h, tokenizer._h, 0))

    @classmethod
    def from_dir(cls, ctx, directory, compute="f32", model_type=None, load_weights=True):
        """Model::new_pretrained (model.rs:68-174) from a sentence-transformers model directory
        (pcv_model_create_from_dir).  load_weights=False: the caller hands the checkpoint tensors to
        `load_hf_tensor` and then calls `check_loaded` (formats other than model.safetensors)."""
        from .tokenizer import BertTokenizer

        self = cls.__new__(cls)
        self.ctx = ctx
        self.model_type = model_type if model_type is not None else SentenceEmbeddingsModelType.AllMiniLmL6V2
        self._h = C.c_void_p()
        try:
            _ffi.check(_ffi.lib().pcv_model_create_from_dir(ctx.handle, str(directory).encode(), _COMPUTE[compute],
                                                            1 if load_weights else 0, C.byref(self._h)))
        except _ffi.PcvError as e:
            raise ModelError(str(e)) from e
        ctx._register(self)
        self.desc = _ffi.ModelDesc()
        pad = C.c_int64()
        _ffi.check(_ffi.lib().pcv_model_get_desc(self._h, C.byref(self.desc), C.byref(pad)))
        self.pad_token_id = pad.value
        th = C.c_void_p()
        _ffi.check(_ffi.lib().pcv_model_tokenizer(self._h, C.byref(th)))
        self.tokenizer = BertTokenizer._borrowed(th.value)  # owned by the model handle
        return self

    def load_hf_tensor(self, name, array):
        """One checkpoint tensor under its Hugging Face / rust-bert name (pcv_model_load_hf_tensor)."""
        a = np.ascontiguousarray(array, dtype=np.float32).reshape(-1)
        try:
            _ffi.check(_ffi.lib().pcv_model_load_hf_tensor(self._handle, name.encode(), _ffi.f32p(a), a.size))
        except _ffi.PcvError as e:
            raise ModelError(str(e)) from e

    def check_loaded(self):
        try:
            _ffi.check(_ffi.lib().pcv_model_check_loaded(self._handle))
        except _ffi.PcvError as e:
            raise ModelError(str(e)) from e

    @property
    def output_dim(self):
        n = C.c_int()
        _ffi.check(_ffi.lib().pcv_model_output_dim(self._handle, C.byref(n)))
        return n.value

    # tokenize.rs:9-57
    def generate_token_tensors(self, token_ids):
        """Right-pad to the batch maximum with the pad id; mask = (id != pad) as int64."""
        max_len = max((len(t) for t in token_ids), default=0)
        ids = np.full((len(token_ids), max_len), self.pad_token_id, dtype=np.int64)
        for i, t in enumerate(token_ids):
            ids[i, : len(t)] = t
        mask = (ids != self.pad_token_id).astype(np.int64)
        return ids, mask

    # model.rs:181-190 + worker.rs:78-106
    def encode_tokens(self, tokens_ids, tokens_masks):
        ids = np.ascontiguousarray(tokens_ids, dtype=np.int64)
        mask = np.ascontiguousarray(tokens_masks, dtype=np.int64)
        if ids.ndim != 2 or ids.shape != mask.shape:
            raise ValueError("tokens_ids / tokens_masks must be [B, L] and alike")
        out = np.empty((ids.shape[0], self.output_dim), dtype=np.float32)
        try:
            _ffi.check(_ffi.lib().pcv_model_encode_tokens(self._handle, _ffi.i64p(ids), _ffi.i64p(mask), ids.shape[0],
                                                          ids.shape[1], _ffi.f32p(out)))
        except _ffi.PcvError as e:
            raise ModelError(str(e)) from e
        return out

    def encode_tokens_device(self, tokens_ids, tokens_masks, d_out, async_=False):
        ids = np.ascontiguousarray(tokens_ids, dtype=np.int64)
        mask = np.ascontiguousarray(tokens_masks, dtype=np.int64)
        _ffi.check(_ffi.lib().pcv_model_encode_tokens_device(self._handle, _ffi.i64p(ids), _ffi.i64p(mask), ids.shape[0],
                                                             ids.shape[1], C.c_void_p(d_out), 1 if async_ else 0))

    def tokenize(self, inputs):
        """Model::tokenize (tokenize.rs:60-77): encode_list(inputs, max_seq_length, LongestFirst, 0),
        then generate_token_tensors."""
        if self.tokenizer is None:
            raise ModelError("this Model was built without a tokenizer (pass tokenizer=BertTokenizer(vocab.txt))")
        inputs = list(inputs)
        if hasattr(self.tokenizer, "encode_batch_ids"):  # threaded C++ batch path; same ids as encode_list
            ids, lens = self.tokenizer.encode_batch_ids(inputs, self.desc.max_seq_length, self.pad_token_id,
                                                        n_threads=_tokenizer_threads())
            ids = np.ascontiguousarray(ids[:, : int(lens.max()) if len(inputs) else 0])
            return ids, (ids != self.pad_token_id).astype(np.int64)
        enc = self.tokenizer.encode_list(inputs, self.desc.max_seq_length)
        return self.generate_token_tensors([e.token_ids for e in enc])

    @staticmethod
    def _c_strings(texts):
        raw = [t.encode("utf-8") for t in texts]
        n = len(raw)
        return raw, (C.c_char_p * max(n, 1))(*raw), (C.c_size_t * max(n, 1))(*[len(b) for b in raw])

    def encode(self, inputs):
        """Model::encode (model.rs:176-179): tokenize + forward in the library (pcv_model_encode_text)."""
        inputs = list(inputs)
        out = np.empty((len(inputs), self.output_dim), dtype=np.float32)
        if not inputs:
            return out
        raw, arr, nb = self._c_strings(inputs)
        try:
            _ffi.check(_ffi.lib().pcv_model_encode_text(self._handle, arr, nb, len(raw), _ffi.f32p(out)))
        except _ffi.PcvError as e:
            raise ModelError(str(e)) from e
        return out

    def highlight(self, query, documents, chunk_size=0, chunk_overlap=-1):
        """Model::highlight (highlight.rs:23-165), computed by pcv_model_highlight: for each document the text
        of the chunk that matches the query best, None when the document is too short for a chunk.
        chunk_size / chunk_overlap default to the CHUNK_SIZE / CHUNK_OVERLAP environment (20 / 4)."""
        documents = list(documents)
        raw, arr, nb = self._c_strings(documents)
        n = len(documents)
        begin = np.full(max(n, 1), -1, dtype=np.int64)
        end = np.full(max(n, 1), -1, dtype=np.int64)
        q = query.encode("utf-8")
        try:
            _ffi.check(_ffi.lib().pcv_model_highlight(self._handle, q, len(q), arr, nb, n, int(chunk_size), int(chunk_overlap),
                                                      _ffi.i64p(begin), _ffi.i64p(end)))
        except _ffi.PcvError as e:
            raise ModelError(str(e)) from e
        return [None if begin[i] < 0 else raw[i][begin[i]:end[i]].decode("utf-8") for i in range(n)]

    # ---- weights / diagnostics ----------------------------------------------------------------
    def get_tensor(self, name):
        n = C.c_int64()
        _ffi.check(_ffi.lib().pcv_model_get_tensor(self._handle, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        _ffi.check(_ffi.lib().pcv_model_get_tensor(self._handle, name.encode(), _ffi.f32p(out), out.size, C.byref(n)))
        return out

    def set_tensor(self, name, data):
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        _ffi.check(_ffi.lib().pcv_model_set_tensor(self._handle, name.encode(), _ffi.f32p(a), a.size))

    def tensor_names(self):
        d = self.desc
        names = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
                 "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
        per = ["attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
               "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"]
        for i in range(d.layers):
            for p in per:
                names += [f"encoder.layer.{i}.{p}.weight", f"encoder.layer.{i}.{p}.bias"]
        if d.dense_out > 0:
            names += ["dense.linear.weight", "dense.linear.bias"]
        return names

    def state_dict(self):
        d = self.desc
        shapes = {}
        H, F = d.hidden, d.intermediate
        out = {}
        for n in self.tensor_names():
            a = self.get_tensor(n)
            if n.endswith("word_embeddings.weight"):
                a = a.reshape(d.vocab_size, H)
            elif n.endswith("position_embeddings.weight"):
                a = a.reshape(d.max_positions, H)
            elif n.endswith("token_type_embeddings.weight"):
                a = a.reshape(d.type_vocab, H)
            elif n.endswith("intermediate.dense.weight"):
                a = a.reshape(F, H)
            elif n.endswith("output.dense.weight") and "attention" not in n:
                a = a.reshape(H, F)
            elif n == "dense.linear.weight":
                a = a.reshape(d.dense_out, H)
            elif n.endswith(".weight") and "LayerNorm" not in n:
                a = a.reshape(H, H)
            out[n] = a
        return out

    def load_state_dict(self, tensors):
        for n in self.tensor_names():
            self.set_tensor(n, tensors[n])

    def debug_hidden(self, layer, B, L):
        out = np.empty((B, L, self.desc.hidden), dtype=np.float32)
        _ffi.check(_ffi.lib().pcv_model_debug_hidden(self._handle, int(layer), _ffi.f32p(out), out.size))
        return out

    def last_stats(self):
        st = _ffi.EncodeStats()
        _ffi.check(_ffi.lib().pcv_model_last_stats(self._handle, C.byref(st)))
        return {"total_ms": st.total_ms, "flops": st.flops, "batch": st.batch, "seq_len": st.seq_len}

    @property
    def _handle(self):
        if not self._h:
            raise RuntimeError("model already closed")
        return self._h

    def close(self):
        if self._h:
            if self.ctx._h:  # a context that is gone took its handles with it
                _ffi.lib().pcv_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
