"""GPU, two fresh processes: the device protocol of a sharded search as configs[3] runs it — every rank holds its rows of
the corpus in HBM, `search_device_begin -> all-gather of B*k+1 records -> merge_topk_flagged -> search_device_end`, repeats
decided from the exchanged overflow record — with both ranks on GPU 0 and the exchange staged through the host over gloo
(the one-GPU box has no second device for RCCL to talk to; the collective itself is torch's).  Covers a step that every
rank repeats because ONE rank's candidate lists overflowed, and one repeated because a speculative start threshold failed
on one rank; results against the oracle on the whole corpus."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PCV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCV_ROOT"], "tests"))
import torch, torch.distributed as dist          # torch first: one HIP runtime for it and the library
import oracle_ffi
import perceive_amd as pa

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = oracle_ffi.load()
N, D, k = 800_000, 128, 10
rng = np.random.default_rng(77)                    # the same corpus and queries on both ranks
m = rng.standard_normal((N, D)).astype(np.float32)
q = rng.standard_normal((6, D)).astype(np.float32)
lo, hi = pa.shard_bounds(N, rank, world)
# query 0's ten best rows are SEED rows of rank 0's shard, one per seed group: rank 0's guess for it must fail
nblocks, shift = (hi - lo + 31) // 32, 0
while (512 << (shift + 1)) <= nblocks:
    shift += 1
seed_rows = [((7 + 31 * j) << shift) * 32 + j for j in range(k)]
assert len({((7 + 31 * j) * 32 + j) % k for j in range(k)}) == k and max(seed_rows) < N // 2
for j in range(k):
    m[seed_rows[j]] = q[0] + (0.02 + 0.01 * j) * rng.standard_normal(D).astype(np.float32)
m[N // 2 + 5] = m[1234]                             # a tie across the shard boundary -> lower global position first
q[2] = m[1234]

ctx = pa.Context(0)
s = pa.Searcher(ctx, D, "cosine")
s.add_rows(1, m[lo:hi], (np.arange(lo, hi) * 10 + 1))  # ids are not positions
s.finalize()
s.set_shard_offset(lo)
s.set_kernel("mfma")

calls = []
def gather(gathered, local):                        # device buffers staged through the host: gloo has no device all-gather
    calls.append(local.numel())
    h = local.cpu()
    out = torch.empty(gathered.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, h)
    gathered.copy_(out)

ss = pa.ShardedSearcher(dist, "cosine", D, searcher=s, ctx=ctx, device=True, all_gather=gather)
opos, osc, _ = orc.topk(q, m, k)

# 1. the failed guess on rank 0: one repeat on BOTH ranks, exact result
ids, sc, cnt = ss.search_vectors(None, k, q)
assert len(calls) == 2 and calls[0] == (6 * k + 1) * 24, calls
assert (ids == opos * 10 + 1).all(), (rank, ids[0], opos[0])
assert np.abs(sc - osc.astype(np.float32)).max() < 1e-6 and (cnt == k).all()
assert set(ids[0]) == {r * 10 + 1 for r in seed_rows}
assert list(ids[2][:2]) == [12341, (N // 2 + 5) * 10 + 1]
assert s.last_stats()["speculation_reruns"] == 0    # the repeat ran without a guess on either rank (the last attempt's statistics)

# 2. fresh queries: no repeat
del calls[:]
q2 = rng.standard_normal((64, D)).astype(np.float32)
ids2, sc2, _ = ss.search_vectors(None, k, q2)
assert len(calls) == 1, calls
o2, s2, _ = orc.topk(q2, m, k)
assert (ids2 == o2 * 10 + 1).all()

# 3. rank 1's candidate lists are too short: its pass overflows, the flag travels with the hits, both ranks repeat
if rank == 1:
    s.set_candidate_capacity(16)
del calls[:]
q3 = rng.standard_normal((64, D)).astype(np.float32)
s.set_tuning(32)                                    # no guesses here: the repeat is the overflow's alone
ids3, sc3, _ = ss.search_vectors(None, k, q3)
assert len(calls) >= 2, calls
o3, s3, _ = orc.topk(q3, m, k)
assert (ids3 == o3 * 10 + 1).all()
assert np.abs(sc3 - s3.astype(np.float32)).max() < 1e-6

dist.barrier()
s.close(); ctx.close()
dist.destroy_process_group()
print("rank", rank, "ok", flush=True)
"""


WORKER_E2E = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PCV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PCV_ROOT"], "tests"))
import torch, torch.distributed as dist          # torch first: one HIP runtime for it and the library
import oracle_ffi
import perceive_amd as pa

# BASELINE configs[4] in small, on two ranks: every rank encodes ITS half of the documents and leaves the embeddings on its
# GPU (encode_tokens_device into its slot of the batch's buffer), an all-gather completes the buffer on every rank (staged
# through the host over gloo here: both ranks sit on GPU 0), and the sharded search reads its queries from that DEVICE buffer
# (search_device_begin_dq) — the embeddings never pass through host memory on the product path.
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = oracle_ffi.load()
desc = dict(vocab=500, hidden=128, layers=2, heads=4, inter=256, max_pos=64, eps=1e-12, pooling=0, normalize=1)
ctx = pa.Context(0)
m = pa.Model(ctx, pa.make_desc(500, 128, 2, 4, 256, 64), synthetic_seed=5)     # the same seeded weights on both ranks
rng = np.random.default_rng(3)
docs, L, D, k = 12, 40, 128, 10
lens = rng.integers(5, L + 1, docs); lens[0] = L
toks = [list(rng.integers(1, 500, int(n))) for n in lens]
ids, mask = m.generate_token_tensors(toks)
N = 300_000
corpus = rng.standard_normal((N, D)).astype(np.float32)
lo, hi = pa.shard_bounds(N, rank, world)
s = pa.Searcher(ctx, D, "cosine")
s.add_rows(1, corpus[lo:hi], np.arange(lo, hi) * 3 + 2)
s.finalize()
s.set_shard_offset(lo)

d0, d1 = docs * rank // world, docs * (rank + 1) // world
emb = torch.zeros((docs, D), dtype=torch.float32, device="cuda")               # the batch's embeddings, on the device
m.encode_tokens_device(ids[d0:d1], mask[d0:d1], emb.data_ptr() + d0 * D * 4)   # this rank's documents into their slot
ctx.synchronize()
mine = emb[d0:d1].cpu()                                                         # (the exchange itself: gloo has no device all-gather)
parts = [torch.empty((docs * (r + 1) // world - docs * r // world, D), dtype=torch.float32) for r in range(world)]
dist.all_gather(parts, mine)
emb.copy_(torch.cat(parts))
torch.cuda.synchronize()

def gather(gathered, local):
    h = local.cpu()
    out = torch.empty(gathered.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, h)
    gathered.copy_(out)

ss = pa.ShardedSearcher(dist, "cosine", D, searcher=s, ctx=ctx, device=True, all_gather=gather)
got, sc, cnt = ss.search_device_queries(None, k, emb.data_ptr(), docs)

oemb, _ = orc.encode_tokens(desc, m.state_dict(), ids, mask)                    # the oracle: all documents, whole corpus
assert np.abs(emb.cpu().numpy() - oemb).max() < 1e-4
opos, osc, _ = orc.topk(emb.cpu().numpy(), corpus, k)                           # ranking of the GPU's own embeddings: bit-exact ids
assert (got == opos * 3 + 2).all() and (cnt == k).all()
assert np.abs(sc - osc.astype(np.float32)).max() < 1e-6
opos2, _, _ = orc.topk(oemb, corpus, k)                                         # ... and of the oracle's embeddings: the same documents
assert (np.sort(got, 1) == np.sort(opos2 * 3 + 2, 1)).mean() > 0.98
# the same search from host queries gives the same answer
got_h, sc_h, _ = ss.search_vectors(None, k, emb.cpu().numpy())
assert (got_h == got).all() and (sc_h == sc).all()
# 200 queries in ONE pass among the ranks: only after the host has said that every rank keeps the int8 copy of its rows
q200 = torch.from_numpy(rng.standard_normal((200, D)).astype(np.float32)).cuda()
torch.cuda.synchronize()
try:
    ss.search_device_queries(None, k, q200.data_ptr(), 200)
    raise SystemExit("a 200-query pass among ranks must be refused until the host allows it")
except pa.PcvError as e:
    assert e.status == 3, e
assert s.last_stats()["screening_copy"] == 2
s.allow_wide_sharded_pass(True)
got2, sc2, cnt2 = ss.search_device_queries(None, k, q200.data_ptr(), 200)
o2, os2, _ = orc.topk(q200.cpu().numpy(), corpus, k)
assert (got2 == o2 * 3 + 2).all() and (cnt2 == k).all()
assert np.abs(sc2 - os2.astype(np.float32)).max() < 1e-6
assert s.last_stats()["scan_launches"] == 1
s.allow_wide_sharded_pass(False)
dist.barrier()
m.close(); s.close(); ctx.close()
dist.destroy_process_group()
print("rank", rank, "ok", flush=True)
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_processes_encode_gather_search_with_device_queries(tmp_path):
    _run_two(tmp_path, WORKER_E2E)


def test_two_processes_share_a_corpus_through_the_device_protocol(tmp_path):
    _run_two(tmp_path, WORKER)


def _run_two(tmp_path, worker):
    script = tmp_path / "worker.py"
    script.write_text(worker)
    env = dict(os.environ, PCV_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2",
               OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            p.kill()  # the exact child this test started
            outs.append(p.communicate()[0] + "\n(timed out)")
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-4000:]}"
        assert f"rank {r} ok" in o
