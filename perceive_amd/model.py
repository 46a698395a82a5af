"""Host-side mirror of perceive-core's model module (crates/perceive-core/model.rs,
model/{configs,tokenize,worker}.rs) over the C ABI.

`Model.encode_tokens` is the hot path (worker.rs:78-106) and runs on the GPU.  The WordPiece
tokenizer (rust_tokenizers) and the pretrained checkpoints are not available offline, so a Model is
built from a description + a flat weight file or seeded synthetic weights; `generate_token_tensors`
restates the padding/mask layout of tokenize.rs:9-57 for callers that bring their own token ids.
"""
import ctypes as C
import enum
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


def minilm_l6_desc():
    d = _ffi.ModelDesc()
    _ffi.lib().pcv_model_desc_minilm_l6(C.byref(d))
    return d


def make_desc(vocab_size, hidden, layers, heads, intermediate, max_positions, type_vocab=2, layer_norm_eps=1e-12,
              pooling="mean", normalize=True, dense_out=0, dense_activation="identity", max_seq_length=256):
    pools = {"mean": _ffi.POOL_MEAN, "cls": _ffi.POOL_CLS, "max": _ffi.POOL_MAX, "mean_sqrt_len": _ffi.POOL_MEAN_SQRT_LEN}
    acts = {"identity": _ffi.ACT_IDENTITY, "tanh": _ffi.ACT_TANH}
    return _ffi.ModelDesc(vocab_size, hidden, layers, heads, intermediate, max_positions, type_vocab, layer_norm_eps,
                          pools[pooling], 1 if normalize else 0, dense_out, acts[dense_activation], max_seq_length,
                          _ffi.COMPUTE_F32)


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
                 model_type=SentenceEmbeddingsModelType.AllMiniLmL6V2, pad_token_id=0):
        self.ctx = ctx
        self.model_type = model_type
        self.desc = desc if desc is not None else minilm_l6_desc()
        self.pad_token_id = pad_token_id
        self._h = C.c_void_p()
        wp = weights_path.encode() if weights_path else None
        _ffi.check(_ffi.lib().pcv_model_create(ctx.handle, C.byref(self.desc), wp, int(synthetic_seed), C.byref(self._h)))

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

    def encode(self, inputs):
        """model.rs:176-179.  Needs the WordPiece tokenizer (not built yet)."""
        raise ModelError("Model.encode(&[str]) needs the tokenizer, which is not available offline; "
                         "use encode_tokens(ids, masks)")

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
            _ffi.lib().pcv_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
