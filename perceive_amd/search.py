"""Host-side mirror of perceive-core's search module (crates/perceive-core/search.rs) over the
C ABI.  Same names and argument meaning as the reference; the SQLite `Database` argument of
`Searcher::build` / `rebuild_source` is replaced by the row stream its SQL produces
(search.rs:87-113: `(items.id, source_id, embedding blob)`), because storage is out of scope.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from .context import Context

_METRICS = {"cosine": _ffi.METRIC_COSINE, "dot": _ffi.METRIC_DOT}
_KERNELS = {"auto": _ffi.KERNEL_AUTO, "wave": _ffi.KERNEL_WAVE, "mfma": _ffi.KERNEL_MFMA}
STAGING_SOURCE = -(1 << 63)  # PCV_STAGING_SOURCE


@dataclass(frozen=True)
class SearchItem:
    """search.rs:18-22"""

    id: int
    score: float


def serialize_embedding(embedding):
    """search.rs:288-294 — little-endian f32 bytes, no header."""
    v = np.ascontiguousarray(embedding, dtype=np.float32)
    out = np.empty(v.size * 4, dtype=np.uint8)
    _ffi.check(_ffi.lib().pcv_serialize_embedding(_ffi.f32p(v), v.size, _ffi.u8p(out), out.size))
    return out.tobytes()


def deserialize_embedding(value):
    """search.rs:281-286"""
    b = np.frombuffer(bytes(value), dtype=np.uint8)
    out = np.empty(len(b) // 4, dtype=np.float32)
    n = C.c_size_t()
    _ffi.check(_ffi.lib().pcv_deserialize_embedding(_ffi.u8p(b), b.size, _ffi.f32p(out), out.size, C.byref(n)))
    return out[: n.value]


def _source_filter(sources):
    """(pointer, count, keep-alive) of a source filter for the C ABI: None -> NULL (all sources); a list ->
    exactly those, an empty list matching nothing (search.rs:166).  The pointer of an empty list is still
    non-NULL."""
    if sources is None:
        return None, 0, None
    sa = np.ascontiguousarray(list(sources), dtype=np.int64)
    n = int(sa.size)
    if n == 0:
        sa = np.zeros(1, dtype=np.int64)
    return _ffi.i64p(sa), n, sa


class Searcher:
    """search.rs:29-35.  One exact, GPU-resident index per process; sources are kept apart so
    `search_vector(sources=...)` filters like search.rs:166 and `rebuild_source` replaces one.

    metric="dot"    -> the reference Searcher's convention: score = max(0, 1 - dot/len)
                       (search.rs:269-278), results ascending by score (search.rs:179).
    metric="cosine" -> lib.rs:67-77 cosine, results best (largest) first.
    """

    def __init__(self, ctx: Context, dim: int, metric: str = "cosine"):
        self.ctx = ctx
        self.dim = int(dim)
        self.metric = metric
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().pcv_searcher_create(ctx.handle, self.dim, _METRICS[metric], C.byref(self._h)))
        ctx._register(self)
        # search.rs:31-34: ids hidden after the index was built.  Kept, and (like the reference's
        # search_vector) not consulted when searching.
        self.hidden = set()

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def build(cls, ctx, rows, dim, metric="cosine"):
        """Searcher::build (search.rs:38-56).  `rows` yields (item_id, source_id, embedding) with the
        embedding as a blob (bytes) or a float array — what the query at search.rs:87-113 returns."""
        s = cls(ctx, dim, metric)
        s._insert(rows)
        s.finalize()
        return s

    def rebuild_source(self, rows, source_id):
        """Searcher::rebuild_source (search.rs:58-79): rows of other sources are ignored
        (search.rs:106-109); an empty replacement leaves the source absent."""
        # search.rs:57-79 builds the new index first and swaps it in only once it exists: the replacement is staged
        # under PCV_STAGING_SOURCE, so that a bad row leaves the old rows of the source in place
        lib, staging = _ffi.lib(), STAGING_SOURCE
        _ffi.check(lib.pcv_searcher_clear_source(self._handle, staging))
        try:
            self._insert((r for r in rows if int(r[1]) == int(source_id)), into=staging)
            self.finalize()
        except Exception:
            lib.pcv_searcher_clear_source(self._handle, staging)
            lib.pcv_searcher_finalize(self._handle)
            raise
        _ffi.check(lib.pcv_searcher_replace_source(self._handle, staging, int(source_id)))
        self.finalize()

    def _insert(self, rows, into=None):
        by_source = {}
        for item_id, source_id, emb in rows:
            v = deserialize_embedding(emb) if isinstance(emb, (bytes, bytearray, memoryview)) else np.asarray(
                emb, dtype=np.float32
            )
            if v.shape != (self.dim,):
                raise ValueError(f"embedding of item {item_id} has shape {v.shape}, index is {self.dim}-d")
            ids, vecs = by_source.setdefault(int(source_id) if into is None else into, ([], []))
            ids.append(int(item_id))
            vecs.append(v)
        for source_id, (ids, vecs) in by_source.items():
            self.add_rows(source_id, np.stack(vecs), np.asarray(ids, dtype=np.int64))

    def add_rows(self, source_id, rows, ids=None):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}]")
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            if ids.shape != (rows.shape[0],):
                raise ValueError("ids must be [n]")
            idp = _ffi.i64p(ids)
        _ffi.check(_ffi.lib().pcv_searcher_add_rows(self._handle, int(source_id), idp, _ffi.f32p(rows), rows.shape[0]))

    def add_blobs(self, source_id, blobs: bytes, n, ids=None):
        b = np.frombuffer(blobs, dtype=np.uint8)
        if b.size != n * self.dim * 4:
            raise ValueError("blob bytes do not match n*dim*4")
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            idp = _ffi.i64p(ids)
        _ffi.check(_ffi.lib().pcv_searcher_add_blobs(self._handle, int(source_id), idp, _ffi.u8p(b), int(n)))

    def reserve(self, source_id, n_rows):
        """Announce that `n_rows` more rows are about to be added to `source_id` (one device segment for them)."""
        _ffi.check(_ffi.lib().pcv_searcher_reserve(self._handle, int(source_id), int(n_rows)))

    def add_synthetic(self, source_id, n, seed, first_row=0, normalize=False, n_clusters=0, noise=0.0, amplitude=None):
        """Rows generated on the device.  n_clusters > 0: clustered rows (centroid/sqrt(dim) + noise * row);
        amplitude=(lo, hi): un-normalised rows a(row) * row with a uniform in [lo, hi) (norms spread: dot-metric corpora)."""
        if amplitude is not None:
            _ffi.check(_ffi.lib().pcv_searcher_add_synthetic_scaled(self._handle, int(source_id), int(n), int(seed), int(first_row),
                                                                  float(amplitude[0]), float(amplitude[1])))
            return
        _ffi.check(
            _ffi.lib().pcv_searcher_add_synthetic_clustered(
                self._handle, int(source_id), int(n), int(seed), int(first_row), 1 if normalize else 0,
                int(n_clusters), float(noise)
            )
        )

    def finalize(self):
        _ffi.check(_ffi.lib().pcv_searcher_finalize(self._handle))

    # ---- queries ------------------------------------------------------------------------------
    def search_vector(self, sources, num_results, vector):
        """Searcher::search_vector (search.rs:157-182)."""
        ids, scores, counts = self.search_vectors(sources, num_results, np.asarray(vector, dtype=np.float32)[None, :])
        return [SearchItem(int(ids[0, j]), float(scores[0, j])) for j in range(int(counts[0]))]

    def search_vectors(self, sources, num_results, vectors):
        """Batched search_vector: vectors [B, dim] -> (ids [B,k] int64, scores [B,k] f32, counts [B])."""
        q = np.ascontiguousarray(vectors, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"vectors must be [B, {self.dim}]")
        B, k = q.shape[0], int(num_results)
        ids = np.full((B, k), -1, dtype=np.int64)
        scores = np.full((B, k), np.nan, dtype=np.float32)
        counts = np.zeros(B, dtype=np.int32)
        src, nsrc, _keep = _source_filter(sources)
        _ffi.check(
            _ffi.lib().pcv_searcher_search(
                self._handle, _ffi.f32p(q), B, src, nsrc, k, _ffi.i64p(ids), _ffi.f32p(scores),
                _ffi.i32p(counts),
            )
        )
        return ids, scores, counts

    def search(self, model, sources, num_results, query):
        """Searcher::search (search.rs:184-193)."""
        return self.search_vector(sources, num_results, encode_query(model, query))

    def search_device(self, sources, num_results, vectors, d_out, async_=False):
        """Per-shard exact top-k left on the device: d_out = device pointer to [B][k] pcv_hit."""
        q = np.ascontiguousarray(vectors, dtype=np.float32)
        src, nsrc, _keep = _source_filter(sources)
        _ffi.check(
            _ffi.lib().pcv_searcher_search_device(
                self._handle, _ffi.f32p(q), q.shape[0], src, nsrc, int(num_results), C.c_void_p(d_out),
                1 if async_ else 0,
            )
        )

    def search_device_begin(self, sources, num_results, vectors, d_out):
        """Queue the per-shard pass without waiting (pcv_searcher_search_device_begin): d_out receives
        B*k hits + one overflow record.  Raises PcvError(status 3) if it needs more than one pass."""
        q = np.ascontiguousarray(vectors, dtype=np.float32)
        src, nsrc, _keep = _source_filter(sources)
        _ffi.check(
            _ffi.lib().pcv_searcher_search_device_begin(
                self._handle, _ffi.f32p(q), q.shape[0], src, nsrc, int(num_results), C.c_void_p(d_out)
            )
        )

    def search_device_begin_dq(self, sources, num_results, d_queries, n_queries, d_out):
        """The same with the queries in device memory (pcv_searcher_search_device_begin_dq): `d_queries` is the address of
        [n_queries][dim] f32 on this searcher's device, e.g. what Model.encode_tokens_device left there."""
        src, nsrc, _keep = _source_filter(sources)
        _ffi.check(
            _ffi.lib().pcv_searcher_search_device_begin_dq(
                self._handle, C.c_void_p(d_queries), int(n_queries), src, nsrc, int(num_results), C.c_void_p(d_out)
            )
        )

    def search_device_end(self):
        """Wait for the queued pass and book its statistics; True if the pass has to be repeated (a candidate list
        overflowed, or a speculative start threshold did not hold)."""
        over = C.c_int()
        _ffi.check(_ffi.lib().pcv_searcher_search_device_end(self._handle, C.byref(over)))
        return bool(over.value)

    def repeat_without_guess(self):
        """A sharded step is repeated because some rank's pass was incomplete: this rank's repeat runs without a
        speculative start threshold too (pcv_searcher_repeat_without_guess)."""
        _ffi.check(_ffi.lib().pcv_searcher_repeat_without_guess(self._handle))

    def search_sharded(self, comm, sources, num_results, vectors):
        """Collective exact top-k over every rank's shard (pcv_searcher_search_sharded): local pass,
        RCCL all-gather of the hit lists, merge.  Returns (ids[B,k], scores[B,k], counts[B])."""
        q = np.ascontiguousarray(vectors, dtype=np.float32)
        B, k = q.shape[0], int(num_results)
        src, nsrc, _keep = _source_filter(sources)
        ids = np.full((B, k), -1, dtype=np.int64)
        scores = np.full((B, k), np.nan, dtype=np.float32)
        counts = np.zeros(B, dtype=np.int32)
        _ffi.check(
            _ffi.lib().pcv_searcher_search_sharded(
                self._handle, comm._handle, _ffi.f32p(q), B, src, nsrc, k, _ffi.i64p(ids), _ffi.f32p(scores),
                _ffi.i32p(counts),
            )
        )
        return ids, scores, counts

    def search_sharded_dq(self, comm, sources, num_results, d_queries, n_queries):
        """search_sharded with the queries in device memory, the same on every rank (pcv_searcher_search_sharded_dq)."""
        B, k = int(n_queries), int(num_results)
        src, nsrc, _keep = _source_filter(sources)
        ids = np.full((B, k), -1, dtype=np.int64)
        scores = np.full((B, k), np.nan, dtype=np.float32)
        counts = np.zeros(B, dtype=np.int32)
        _ffi.check(
            _ffi.lib().pcv_searcher_search_sharded_dq(
                self._handle, comm._handle, C.c_void_p(d_queries), B, src, nsrc, k, _ffi.i64p(ids), _ffi.f32p(scores),
                _ffi.i32p(counts),
            )
        )
        return ids, scores, counts

    # ---- introspection ------------------------------------------------------------------------
    def set_kernel(self, kernel="auto"):
        _ffi.check(_ffi.lib().pcv_searcher_set_kernel(self._handle, _KERNELS[kernel]))

    def allow_wide_sharded_pass(self, on=True):
        """Every rank's searcher keeps the int8 copy of all its rows (the host has checked): a pass among ranks may take 256
        queries (pcv_searcher_allow_wide_sharded_pass)."""
        _ffi.check(_ffi.lib().pcv_searcher_allow_wide_sharded_pass(self._handle, 1 if on else 0))

    def wait_background(self):
        """Wait for a mid copy that AUTO is building beside the searches (pcv_searcher_wait_background)."""
        _ffi.check(_ffi.lib().pcv_searcher_wait_background(self._handle))

    def set_candidate_capacity(self, n_candidates):
        """Initial rows per query of a pass's candidate lists (tuning; a pass that needs more repeats itself)."""
        _ffi.check(_ffi.lib().pcv_searcher_set_candidate_capacity(self._handle, int(n_candidates)))

    def set_tuning(self, flags=0, fail_copy_alloc=False):
        """Diagnostic / comparison switches (pcv_searcher_set_tuning): `flags` as PCV_SCAN_FLAGS (csrc/scan.h), e.g. 32 = no
        speculative start threshold; fail_copy_alloc: screening-copy allocations fail while set (tests)."""
        _ffi.check(_ffi.lib().pcv_searcher_set_tuning(self._handle, (int(flags) & 0x3FFFFFFF) | ((1 << 30) if fail_copy_alloc else 0)))

    def set_screening_copy(self, mode="auto"):
        """"off" | "bf16" | "int8" | "auto" (= int8): keep a narrow copy of the scaled rows next to the f32 rows so that
        the coarse screen streams a half / a quarter of the bytes (pcv_searcher_set_screening_copy); built at the next
        finalize.  Results do not depend on it."""
        _ffi.check(_ffi.lib().pcv_searcher_set_screening_copy(self._handle, {"off": 0, "bf16": 1, "on": 1, "auto": 2, "int8": 3}[mode]))

    def set_mid_copy(self, mode="auto"):
        """"off" | "auto" | "on": the row-major 16-bit copy the fine screen reads in front of the f32 rows
        (pcv_searcher_set_mid_copy); results do not depend on it."""
        _ffi.check(_ffi.lib().pcv_searcher_set_mid_copy(self._handle, {"off": 0, "auto": 1, "on": 2}[mode]))

    def set_shard_offset(self, first_global_pos):
        _ffi.check(_ffi.lib().pcv_searcher_set_shard_offset(self._handle, int(first_global_pos)))

    @property
    def num_rows(self):
        n = C.c_int64()
        _ffi.check(_ffi.lib().pcv_searcher_num_rows(self._handle, C.byref(n)))
        return n.value

    @property
    def num_segments(self):
        n = C.c_int()
        _ffi.check(_ffi.lib().pcv_searcher_num_segments(self._handle, C.byref(n)))
        return n.value

    @property
    def source_ids(self):
        n = C.c_int()
        _ffi.check(_ffi.lib().pcv_searcher_num_sources(self._handle, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int64)
        _ffi.check(_ffi.lib().pcv_searcher_source_ids(self._handle, _ffi.i64p(out), out.size))
        return [int(x) for x in out[: n.value]]

    def source_num_rows(self, source_id):
        n = C.c_int64()
        _ffi.check(_ffi.lib().pcv_searcher_source_num_rows(self._handle, int(source_id), C.byref(n)))
        return n.value

    def get_rows(self, positions):
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        rows = np.empty((pos.size, self.dim), dtype=np.float32)
        ids = np.empty(pos.size, dtype=np.int64)
        _ffi.check(_ffi.lib().pcv_searcher_get_rows(self._handle, _ffi.i64p(pos), pos.size, _ffi.f32p(rows), _ffi.i64p(ids)))
        return rows, ids

    def last_stats(self):
        st = _ffi.ScanStats()
        _ffi.check(_ffi.lib().pcv_searcher_last_stats(self._handle, C.byref(st)))
        return {f: getattr(st, f) for f, _ in _ffi.ScanStats._fields_}

    @property
    def _handle(self):
        if not self._h:
            raise RuntimeError("searcher already closed")
        return self._h

    def close(self):
        if self._h:
            if self.ctx._h:  # a context that is gone took its handles with it
                _ffi.lib().pcv_searcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_query(model, query):
    """search.rs:262-264"""
    return np.asarray(model.encode([query]))[0]


def merge_topk(ctx, metric, dim, d_lists, n_shards, n_queries, k, flagged=False):
    """Cross-shard merge of all-gathered per-shard lists (device pointer) -> (ids, scores, counts);
    flagged=True: lists made by search_device_begin (one overflow record per shard), returns
    (ids, scores, counts, any_overflow)."""
    ids = np.full((n_queries, k), -1, dtype=np.int64)
    scores = np.full((n_queries, k), np.nan, dtype=np.float32)
    counts = np.zeros(n_queries, dtype=np.int32)
    args = (
        ctx.handle, _METRICS[metric], int(dim), C.c_void_p(d_lists), int(n_shards), int(n_queries), int(k),
        _ffi.i64p(ids), _ffi.f32p(scores), _ffi.i32p(counts),
    )
    if flagged:
        over = C.c_int()
        _ffi.check(_ffi.lib().pcv_merge_topk_flagged(*args, C.byref(over)))
        return ids, scores, counts, bool(over.value)
    _ffi.check(_ffi.lib().pcv_merge_topk(*args))
    return ids, scores, counts


# ---- lib.rs:63-77 -----------------------------------------------------------------------------
def _sim(ctx, a, m, cosine):
    a = np.ascontiguousarray(a, dtype=np.float32)
    m = np.ascontiguousarray(m, dtype=np.float32)
    out = np.empty((a.shape[0], m.shape[0]), dtype=np.float32)
    fn = _ffi.lib().pcv_cosine_similarity if cosine else _ffi.lib().pcv_dot_product
    _ffi.check(fn(ctx.handle, _ffi.f32p(a), a.shape[0], _ffi.f32p(m), m.shape[0], a.shape[1], _ffi.f32p(out)))
    return out


def dot_product(ctx, set1, set2):
    """lib.rs:63-65: set1.matmul(set2.T) -> [len(set1), len(set2)]"""
    return _sim(ctx, set1, set2, False)


def cosine_similarity_single_query(ctx, query, matches):
    """lib.rs:67-71: query [D], matches [N, D] -> [N]"""
    return _sim(ctx, np.asarray(query, dtype=np.float32)[None, :], matches, True)[0]


def cosine_similarity_multi_query(ctx, set1, set2):
    """lib.rs:73-77"""
    return _sim(ctx, set1, set2, True)
