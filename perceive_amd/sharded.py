"""Row-sharded search across the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI; "gloo" on CPU for the protocol tests).

The corpus never moves: rank r owns rows [r*N/G, (r+1)*N/G) with global positions, computes its
exact local top-k, and the only exchange is one all-gather of the [B][k] hit lists (24 B per hit,
15 KB per rank at B=64, k=10), merged identically on every rank — the multi-GPU form of the
reference's per-source `flat_map_iter` + `sort_unstable_by` + `truncate` (search.rs:163-181).
"""
import ctypes as C

import numpy as np

from . import _ffi
from .search import _METRICS, merge_topk

HIT_DTYPE = np.dtype([("score", "<f8"), ("pos", "<i8"), ("id", "<i8")])
HIT_BYTES = HIT_DTYPE.itemsize  # 24 = sizeof(pcv_hit)


def shard_bounds(n_rows, rank, world):
    """Contiguous, near-equal row ranges in rank order (global position order is preserved, so
    ties break the same way as on one GPU)."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def merge_topk_host(metric, dim, lists, n_shards, n_queries, k):
    """Host merge of gathered lists: `lists` is a [n_shards, n_queries, k] HIT_DTYPE array."""
    a = np.ascontiguousarray(lists).view(HIT_DTYPE).reshape(n_shards, n_queries, k)
    ids = np.full((n_queries, k), -1, dtype=np.int64)
    scores = np.full((n_queries, k), np.nan, dtype=np.float32)
    counts = np.zeros(n_queries, dtype=np.int32)
    _ffi.check(
        _ffi.lib().pcv_merge_topk_host(
            _METRICS[metric], int(dim), a.ctypes.data, n_shards, n_queries, k,
            _ffi.i64p(ids), _ffi.f32p(scores), _ffi.i32p(counts),
        )
    )
    return ids, scores, counts


def hip_runtimes_loaded():
    """Paths of the libamdhip64 copies mapped into this process.  More than one (PyTorch bundles its own
    and is loaded beside the system one when it is imported after this library) means torch's streams
    and this library's are unrelated objects: no device-side ordering between them is possible."""
    seen = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if "libamdhip64" in path:
                    seen.add(path)
    except OSError:
        pass
    return sorted(seen)


class NativeComm:
    """A persistent RCCL communicator owned by libperceive_hip.so (pcv_comm_*): the exchange then runs
    on the library's own stream with no PyTorch in the data path.  `unique_id` is the 128-byte id rank 0
    obtained from `NativeComm.unique_id()` and handed to every rank (see `NativeComm.from_dist`)."""

    def __init__(self, ctx, world, rank, unique_id):
        idb = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        h = C.c_void_p()
        _ffi.check(_ffi.lib().pcv_comm_create(ctx.handle, int(world), int(rank), idb, C.byref(h)))
        self._handle, self.ctx, self.world, self.rank = h, ctx, int(world), int(rank)
        ctx._register(self)

    @staticmethod
    def unique_id():
        idb = (C.c_uint8 * 128)()
        _ffi.check(_ffi.lib().pcv_comm_unique_id(idb))
        return bytes(idb)

    @classmethod
    def from_dist(cls, ctx, dist):
        """Bootstrap over an initialised torch.distributed group (any backend): rank 0's id is
        broadcast as a Python object; torch is used for nothing else."""
        box = [cls.unique_id() if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(ctx, dist.get_world_size(), dist.get_rank(), box[0])

    def all_gather(self, d_send, d_recv, bytes_per_rank):
        """ncclAllGather on the library's communicator and stream (pcv_comm_all_gather): device addresses; d_send may be
        this rank's slot of d_recv."""
        _ffi.check(_ffi.lib().pcv_comm_all_gather(self._handle, C.c_void_p(d_send), C.c_void_p(d_recv), int(bytes_per_rank)))

    def close(self):
        """Destroy the communicator.  It lives on the context's device and stream, so it has to go
        first; if the context is already closed the handle is dropped, not destroyed."""
        h, self._handle = getattr(self, "_handle", None), None
        if h and self.ctx._h:
            _ffi.check(_ffi.lib().pcv_comm_destroy(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedSearcher:
    """Wraps this rank's local Searcher.  `dist` is an initialised torch.distributed module.

    device=True : hit lists stay in HBM (torch CUDA byte tensors as the exchange buffers), RCCL
                  all-gather, device merge kernel.
    comm=NativeComm: the same exchange done by the library itself (ncclAllGather on its stream + merge
                  kernel, pcv_searcher_search_sharded); `dist` may then be None.
    device=False: lists are gathered on the host (gloo) and merged by pcv_merge_topk_host;
                  `local_search` may then be any callable (queries, k) -> [B,k] HIT_DTYPE array,
                  which is how the CPU tests drive the protocol without a GPU.
    """

    def __init__(self, dist, metric, dim, searcher=None, ctx=None, device=True, local_search=None, comm=None,
                 all_gather=None):
        self.dist = dist
        self.comm = comm  # NativeComm: exchange inside the library (pcv_searcher_search_sharded)
        self.world = comm.world if comm is not None else dist.get_world_size()
        self.rank = comm.rank if comm is not None else dist.get_rank()
        self.metric, self.dim = metric, int(dim)
        self.searcher, self.ctx = searcher, ctx
        self.device = device
        self.local_search = local_search
        # the collective itself: (gathered, local) -> None; replaceable for rehearsals (e.g. staging
        # device buffers through the host over gloo on a one-GPU box)
        self._all_gather = all_gather or (lambda gathered, local: dist.all_gather_into_tensor(gathered, local))
        self._bufs = {}
        self._adopted = False

    def _buffers(self, B, k):
        import torch

        key = (B, k)
        if key not in self._bufs:
            dev = "cuda" if self.device else "cpu"
            rec = B * k + 1  # + the overflow record of search_device_begin
            self._bufs[key] = (
                torch.empty(rec * HIT_BYTES, dtype=torch.uint8, device=dev),
                torch.empty(self.world * rec * HIT_BYTES, dtype=torch.uint8, device=dev),
            )
        return self._bufs[key]

    def search_vectors(self, sources, num_results, vectors):
        q = np.ascontiguousarray(vectors, dtype=np.float32)
        B, k = q.shape[0], int(num_results)
        if self.comm is not None:
            return self.searcher.search_sharded(self.comm, sources, k, q)
        import torch  # only the torch.distributed exchange needs it (and it must then be imported first)

        local, gathered = self._buffers(B, k)
        n = B * k * HIT_BYTES
        if self.device:
            if self._adopted is False:
                # the library queues on torch's current stream from now on: pass -> all-gather -> merge
                # are ordered on the device and the host waits once per step.  Only sound when torch and
                # the library share one HIP runtime (torch imported first); otherwise keep host syncs.
                if len(hip_runtimes_loaded()) == 1:
                    self.ctx.set_stream(torch.cuda.current_stream().cuda_stream, adopt=True)
                    self._adopted = True
                else:
                    self._adopted = None
            attempts = 8 if self._adopted else 0
            while attempts > 0:
                try:
                    self.searcher.search_device_begin(sources, k, q, local.data_ptr())
                except _ffi.PcvError as e:
                    if e.status != 3:  # PCV_ERR_UNSUPPORTED: needs several passes -> sequential form below
                        raise
                    break
                self._all_gather(gathered, local)
                ids, scores, counts, over = merge_topk(
                    self.ctx, self.metric, self.dim, gathered.data_ptr(), self.world, B, k, flagged=True
                )
                self.searcher.search_device_end()
                if not over:  # the same on every rank: it travelled with the hits
                    return ids, scores, counts
                self.searcher.repeat_without_guess()  # on every rank: failed guesses must not take turns between the ranks
                attempts -= 1
                if attempts == 0:
                    raise RuntimeError("the sharded step is still incomplete on some rank after 8 repeats (candidate lists keep overflowing)")
            self.searcher.search_device(sources, k, q, local.data_ptr())  # returns after its stream drained
            self._all_gather(gathered[: self.world * n], local[:n])
            torch.cuda.current_stream().synchronize()
            return merge_topk(self.ctx, self.metric, self.dim, gathered.data_ptr(), self.world, B, k)
        hits = self.local_search(q, k) if self.local_search else self._local_hits_host(sources, q, k)
        local[:n].copy_(torch.from_numpy(np.ascontiguousarray(hits).view(np.uint8).reshape(-1)))
        self._all_gather(gathered[: self.world * n], local[:n])
        return merge_topk_host(self.metric, self.dim, gathered[: self.world * n].numpy(), self.world, B, k)

    def search_device_queries(self, sources, num_results, d_queries, n_queries):
        """search_vectors with the queries in DEVICE memory (address of [n_queries][dim] f32 on this rank's GPU, the same
        values on every rank): the embeddings a data-parallel encode left on the devices and an all-gather put together
        go into the scan without passing through host memory (BASELINE configs[4]).  One pass: n_queries <= 128, or 256 up to
        384-d after Searcher.allow_wide_sharded_pass on every rank."""
        B, k = int(n_queries), int(num_results)
        if self.comm is not None:
            return self.searcher.search_sharded_dq(self.comm, sources, k, d_queries, B)
        import torch

        assert self.device, "device queries need the device protocol"
        local, gathered = self._buffers(B, k)
        if self._adopted is False:
            if len(hip_runtimes_loaded()) == 1:
                self.ctx.set_stream(torch.cuda.current_stream().cuda_stream, adopt=True)
                self._adopted = True
            else:
                self._adopted = None
        for _ in range(8):
            self.searcher.search_device_begin_dq(sources, k, d_queries, B, local.data_ptr())
            if not self._adopted:
                self.ctx.synchronize()  # (two HIP runtimes in the process: no stream is shared, the host orders the steps)
            self._all_gather(gathered, local)
            ids, scores, counts, over = merge_topk(self.ctx, self.metric, self.dim, gathered.data_ptr(), self.world, B, k, flagged=True)
            self.searcher.search_device_end()
            if not over:
                return ids, scores, counts
            self.searcher.repeat_without_guess()
        raise RuntimeError("the sharded step is still incomplete on some rank after 8 repeats (candidate lists keep overflowing)")

    def _local_hits_host(self, sources, q, k):
        d = self.ctx.alloc(q.shape[0] * k * HIT_BYTES)
        try:
            self.searcher.search_device(sources, k, q, d)
            return self.ctx.to_host(d, q.shape[0] * k * HIT_BYTES).view(HIT_DTYPE).reshape(q.shape[0], k)
        finally:
            self.ctx.free(d)
