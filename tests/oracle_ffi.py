"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by perceive_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_F = C.POINTER(C.c_float)
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int64)
_U8 = C.POINTER(C.c_uint8)


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def _fp(a):
    return a.ctypes.data_as(_F)


def _ip(a):
    return a.ctypes.data_as(_I)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_dot_product.argtypes = [_F, C.c_int, _F, C.c_int64, C.c_int, _F]
        L.orc_cosine_similarity_multi_query.argtypes = [_F, C.c_int, _F, C.c_int64, C.c_int, _F]
        L.orc_cosine_similarity_single_query.argtypes = [_F, _F, C.c_int64, C.c_int, _F]
        L.orc_canonical_score.argtypes = [_F, _F, C.c_int, C.c_int]
        L.orc_canonical_score.restype = C.c_double
        L.orc_topk.argtypes = [_F, _F, C.c_int64, C.c_int, C.c_int, C.c_int, _I, _D]
        L.orc_topk.restype = C.c_int
        L.orc_ndarray_distance.argtypes = [_F, _F, C.c_int]
        L.orc_ndarray_distance.restype = C.c_float
        L.orc_search_vector.argtypes = [_F, _F, _I, _I, C.c_int64, C.c_int, _I, C.c_int, C.c_int, _I, _F]
        L.orc_search_vector.restype = C.c_int
        L.orc_serialize_embedding.argtypes = [_F, C.c_size_t, _U8]
        L.orc_deserialize_embedding.argtypes = [_U8, C.c_size_t, _F]
        L.orc_deserialize_embedding.restype = C.c_size_t
        L.orc_synth_rows.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, _F]
        L.orc_synth_rows_clustered.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, _F]
        L.orc_synth_rows_scaled.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_float, _F]
        L.orc_baseline_scan_fused.argtypes = [_F, C.c_int, _F, C.c_int64, C.c_int, C.c_int, C.c_int, _I, _F]
        L.orc_baseline_scan_fused.restype = C.c_double
        L.orc_baseline_scan_reference_shaped.argtypes = L.orc_baseline_scan_fused.argtypes
        L.orc_baseline_scan_reference_shaped.restype = C.c_double
        L.orc_hardware_threads.restype = C.c_int

    # lib.rs:63-77
    def dot_product(self, a, m):
        a = np.ascontiguousarray(a, np.float32)
        m = np.ascontiguousarray(m, np.float32)
        out = np.empty((a.shape[0], m.shape[0]), np.float32)
        self.lib.orc_dot_product(_fp(a), a.shape[0], _fp(m), m.shape[0], a.shape[1], _fp(out))
        return out

    def cosine_similarity_multi_query(self, a, m):
        a = np.ascontiguousarray(a, np.float32)
        m = np.ascontiguousarray(m, np.float32)
        out = np.empty((a.shape[0], m.shape[0]), np.float32)
        self.lib.orc_cosine_similarity_multi_query(_fp(a), a.shape[0], _fp(m), m.shape[0], a.shape[1], _fp(out))
        return out

    def cosine_similarity_single_query(self, q, m):
        q = np.ascontiguousarray(q, np.float32)
        m = np.ascontiguousarray(m, np.float32)
        out = np.empty(m.shape[0], np.float32)
        self.lib.orc_cosine_similarity_single_query(_fp(q), _fp(m), m.shape[0], m.shape[1], _fp(out))
        return out

    def canonical_score(self, q, x, metric=0):
        q = np.ascontiguousarray(q, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        return self.lib.orc_canonical_score(_fp(q), _fp(x), q.shape[0], metric)

    def topk(self, queries, m, k, metric=0):
        """Exact canonical top-k per query -> (pos [B,k] (-1 padded), score64 [B,k], counts [B])."""
        queries = np.ascontiguousarray(queries, np.float32)
        m = np.ascontiguousarray(m, np.float32)
        B = queries.shape[0]
        pos = np.full((B, k), -1, np.int64)
        sc = np.full((B, k), np.nan, np.float64)
        cnt = np.zeros(B, np.int32)
        for b in range(B):
            cnt[b] = self.lib.orc_topk(
                _fp(queries[b]), _fp(m), m.shape[0], m.shape[1], metric, k, _ip(pos[b]), sc[b].ctypes.data_as(_D)
            )
        return pos, sc, cnt

    def ndarray_distance(self, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        return self.lib.orc_ndarray_distance(_fp(a), _fp(b), a.shape[0])

    def search_vector(self, q, m, ids, source_of_row, sources, k):
        q = np.ascontiguousarray(q, np.float32)
        m = np.ascontiguousarray(m, np.float32)
        ids = np.ascontiguousarray(ids, np.int64)
        sor = np.ascontiguousarray(source_of_row, np.int64)
        src = np.ascontiguousarray(list(sources), np.int64)
        out_ids = np.full(k, -1, np.int64)
        out_d = np.full(k, np.nan, np.float32)
        n = self.lib.orc_search_vector(
            _fp(q), _fp(m), _ip(ids), _ip(sor), m.shape[0], m.shape[1], _ip(src), src.size, k, _ip(out_ids), _fp(out_d)
        )
        return out_ids[:n], out_d[:n]

    def serialize_embedding(self, v):
        v = np.ascontiguousarray(v, np.float32)
        out = np.empty(v.size * 4, np.uint8)
        self.lib.orc_serialize_embedding(_fp(v), v.size, out.ctypes.data_as(_U8))
        return out.tobytes()

    def deserialize_embedding(self, blob):
        b = np.frombuffer(blob, np.uint8)
        out = np.empty(b.size // 4, np.float32)
        n = self.lib.orc_deserialize_embedding(b.ctypes.data_as(_U8), b.size, _fp(out))
        return out[:n]

    def synth_rows(self, seed, first_row, n, D, normalize=False):
        out = np.empty((n, D), np.float32)
        self.lib.orc_synth_rows(seed, first_row, n, D, 1 if normalize else 0, _fp(out))
        return out

    def synth_rows_clustered(self, seed, first_row, n, D, n_clusters, noise, normalize=False):
        out = np.empty((n, D), np.float32)
        self.lib.orc_synth_rows_clustered(seed, first_row, n, D, 1 if normalize else 0, n_clusters, noise, _fp(out))
        return out

    def synth_rows_scaled(self, seed, first_row, n, D, amp_lo, amp_hi):
        out = np.empty((n, D), np.float32)
        self.lib.orc_synth_rows_scaled(seed, first_row, n, D, amp_lo, amp_hi, _fp(out))
        return out

    def baseline_scan(self, queries, m, k, threads, shaped=False):
        queries = np.ascontiguousarray(queries, np.float32)
        m = np.ascontiguousarray(m, np.float32)
        B = queries.shape[0]
        pos = np.empty((B, k), np.int64)
        sc = np.empty((B, k), np.float32)
        fn = self.lib.orc_baseline_scan_reference_shaped if shaped else self.lib.orc_baseline_scan_fused
        secs = fn(_fp(queries), B, _fp(m), m.shape[0], m.shape[1], k, threads, _ip(pos), _fp(sc))
        return secs, pos, sc

    def hardware_threads(self):
        return self.lib.orc_hardware_threads()


_cached = None


def load():
    global _cached
    if _cached is None:
        if not os.path.exists(LIB):
            build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached


# ---- encoder oracle (oracle/encoder.c) -----------------------------------------------------------
class _Desc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("vocab", "hidden", "layers", "heads", "inter", "max_pos", "type_vocab")] + [
        ("eps", C.c_float)] + [(n, C.c_int) for n in ("pooling", "normalize", "dense_out", "dense_act")]


_PP = C.POINTER(_F)


class _Weights(C.Structure):
    _fields_ = [(n, _F) for n in ("word", "pos", "type", "emb_ln_w", "emb_ln_b")] + [
        (n, _PP) for n in ("qw", "qb", "kw", "kb", "vw", "vb", "ow", "ob", "ln1w", "ln1b", "iw", "ib", "fw", "fb",
                           "ln2w", "ln2b")] + [("dense_w", _F), ("dense_b", _F)]


_LAYER_KEYS = {
    "qw": "attention.self.query.weight", "qb": "attention.self.query.bias",
    "kw": "attention.self.key.weight", "kb": "attention.self.key.bias",
    "vw": "attention.self.value.weight", "vb": "attention.self.value.bias",
    "ow": "attention.output.dense.weight", "ob": "attention.output.dense.bias",
    "ln1w": "attention.output.LayerNorm.weight", "ln1b": "attention.output.LayerNorm.bias",
    "iw": "intermediate.dense.weight", "ib": "intermediate.dense.bias",
    "fw": "output.dense.weight", "fb": "output.dense.bias",
    "ln2w": "output.LayerNorm.weight", "ln2b": "output.LayerNorm.bias",
}


def encode_tokens(self, desc, weights, ids, mask, want_hidden=False):
    """desc: dict(vocab, hidden, layers, heads, inter, max_pos, eps, pooling, normalize, dense_out,
    dense_act); weights: dict of HF-named f32 arrays.  Returns (out [B,OD], hidden [(layers+1),B,L,H] | None)."""
    keep = []

    def arr(name):
        a = np.ascontiguousarray(weights[name], np.float32)
        keep.append(a)
        return _fp(a)

    d = _Desc(desc["vocab"], desc["hidden"], desc["layers"], desc["heads"], desc["inter"], desc["max_pos"],
              desc.get("type_vocab", 2), desc.get("eps", 1e-12), desc.get("pooling", 0), desc.get("normalize", 1),
              desc.get("dense_out", 0), desc.get("dense_act", 0))
    w = _Weights()
    w.word = arr("embeddings.word_embeddings.weight")
    w.pos = arr("embeddings.position_embeddings.weight")
    w.type = arr("embeddings.token_type_embeddings.weight")
    w.emb_ln_w = arr("embeddings.LayerNorm.weight")
    w.emb_ln_b = arr("embeddings.LayerNorm.bias")
    for field, key in _LAYER_KEYS.items():
        ptrs = (_F * desc["layers"])(*[arr(f"encoder.layer.{i}.{key}") for i in range(desc["layers"])])
        keep.append(ptrs)
        setattr(w, field, C.cast(ptrs, _PP))
    if desc.get("dense_out", 0) > 0:
        w.dense_w = arr("dense.linear.weight")
        w.dense_b = arr("dense.linear.bias")
    ids = np.ascontiguousarray(ids, np.int64)
    mask = np.ascontiguousarray(mask, np.int64)
    B, L = ids.shape
    od = desc.get("dense_out", 0) or desc["hidden"]
    out = np.empty((B, od), np.float32)
    hidden = np.empty((desc["layers"] + 1, B, L, desc["hidden"]), np.float32) if want_hidden else None
    self.lib.orc_encode_tokens.argtypes = [C.POINTER(_Desc), C.POINTER(_Weights), _I, _I, C.c_int, C.c_int, _F, _F]
    self.lib.orc_encode_tokens.restype = None
    self.lib.orc_encode_tokens(C.byref(d), C.byref(w), _ip(ids), _ip(mask), B, L, _fp(hidden) if want_hidden else None,
                               _fp(out))
    return out, hidden


Oracle.encode_tokens = encode_tokens
