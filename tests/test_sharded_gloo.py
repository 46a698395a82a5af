"""CPU, world_size 2, gloo: the multi-GPU protocol (shard bounds, global positions, all-gather
layout, merge with cross-shard ties) without a GPU.  The per-shard search is played by the oracle;
everything else is the product's host code (perceive_amd/sharded.py + pcv_merge_topk_host)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PCV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCV_ROOT"], "tests"))
import torch, torch.distributed as dist
import oracle_ffi
import perceive_amd as pa

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = oracle_ffi.load()
N, D, B, k = 5000, 64, 7, 10
corpus = orc.synth_rows(5, 0, N, D)
corpus[4000] = corpus[100]            # duplicate across the shard boundary: tie -> lower global position
corpus[4500] = 0.0                    # zero row in the second shard
queries = orc.synth_rows(6, 0, B, D)
queries[3] = corpus[100]
lo, hi = pa.shard_bounds(N, rank, world)

def local_search(q, kk):
    pos, sc, cnt = orc.topk(q, corpus[lo:hi], kk)
    hits = np.zeros((q.shape[0], kk), pa.HIT_DTYPE)
    hits["score"] = sc
    hits["pos"] = np.where(pos >= 0, pos + lo, -1)
    hits["id"] = np.where(pos >= 0, (pos + lo) * 10 + 1, -1)   # ids are not positions
    return hits

ss = pa.ShardedSearcher(dist, "cosine", D, device=False, local_search=local_search)
ids, scores, counts = ss.search_vectors(None, k, queries)
opos, osc, ocnt = orc.topk(queries, corpus, k)
assert (ids == opos * 10 + 1).all(), (rank, ids, opos)
assert np.abs(scores - osc.astype(np.float32)).max() < 1e-7
assert list(ids[3][:2]) == [1001, 40001]
assert (counts == k).all()
# k larger than a shard's valid rows: a tiny third corpus
tiny = corpus[:3]
def tiny_search(q, kk):
    l, h = pa.shard_bounds(3, rank, world)
    pos, sc, cnt = orc.topk(q, tiny[l:h], kk) if h > l else (np.full((q.shape[0], kk), -1), np.full((q.shape[0], kk), np.nan), None)
    hits = np.zeros((q.shape[0], kk), pa.HIT_DTYPE)
    hits["score"] = sc; hits["pos"] = np.where(pos >= 0, pos + l, -1); hits["id"] = hits["pos"]
    return hits
ss2 = pa.ShardedSearcher(dist, "cosine", D, device=False, local_search=tiny_search)
ids2, sc2, cnt2 = ss2.search_vectors(None, 5, queries[:2])
op2, os2, oc2 = orc.topk(queries[:2], tiny, 5)
assert (ids2 == op2).all() and (cnt2 == 3).all()
# a caller-supplied collective (what bench.py's one-GPU rehearsal uses) is the one that gets called
calls = []
def my_gather(gathered, local):
    calls.append(local.numel())
    dist.all_gather_into_tensor(gathered, local)
ss3 = pa.ShardedSearcher(dist, "cosine", D, device=False, local_search=local_search, all_gather=my_gather)
ids3, _, _ = ss3.search_vectors(None, k, queries)
assert (ids3 == ids).all() and calls == [B * k * 24]
assert isinstance(pa.sharded.hip_runtimes_loaded(), list)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_protocol(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PCV_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2",
               OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_shard_bounds_cover_rows_in_order():
    import perceive_amd as pa

    for n in (0, 1, 7, 100_000_000):
        for w in (1, 2, 3, 8):
            b = [pa.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_merge_topk_host_orders_and_truncates():
    import perceive_amd as pa

    lists = np.zeros((2, 1, 3), pa.HIT_DTYPE)
    lists["pos"] = -1
    lists["score"] = np.nan
    lists[0, 0, :2] = [(0.5, 10, 100), (0.25, 11, 110)]
    lists[1, 0, :3] = [(0.5, 7, 70), (0.4, 8, 80), (0.1, 9, 90)]
    ids, sc, cnt = pa.merge_topk_host("cosine", 4, lists, 2, 1, 3)
    assert list(ids[0]) == [70, 100, 80] and cnt[0] == 3  # tie at 0.5 -> lower position (7) first
    ids, sc, cnt = pa.merge_topk_host("dot", 4, lists, 2, 1, 3)
    np.testing.assert_allclose(sc[0], [1 - 0.5 / 4, 1 - 0.5 / 4, 1 - 0.4 / 4])  # search.rs:275
