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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_processes_share_a_corpus_through_the_device_protocol(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
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
