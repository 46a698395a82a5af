#!/usr/bin/env python3
"""End to end (BASELINE configs[4]): encode a batch of token sequences with the all-MiniLM-L6-v2-shaped
encoder (seq_len 256, batch 256, seeded synthetic weights), then use the 256 embeddings as queries of an
exact top-10 scan over the synthetic corpus (one pass of 256 queries with the int8 copy).  Prints one JSON line.

    python tools/bench_e2e.py                                            # one GPU
    python -m torch.distributed.run --nproc-per-node N tools/bench_e2e.py --gpus N

N > 1: the encoder runs as data-parallel replicas (weights replicated, batch/N
documents per rank), the embeddings are all-gathered so that every rank holds the full query tile, and the
corpus is row-sharded exactly as in bench.py (per-shard top-k -> all-gather -> merge).  PCV_BENCH_REHEARSE=1
puts all ranks on GPU 0 over gloo to exercise that path on a one-GPU box."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--pass-queries", type=int, default=128, help="queries per corpus pass (<= 128 at 384-d)")
    ap.add_argument("--compute", default="f32", choices=["f32", "bf16x3", "f16x2"])
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dist = torch = None
    rehearse = os.environ.get("PCV_BENCH_REHEARSE") == "1"
    if world > 1:
        import torch  # first: see DESIGN.md §7 (one HIP runtime per process)
        import torch.distributed as dist

        if rehearse:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import perceive_amd as pa

    ctx = pa.Context(local_rank)
    m = pa.Model(ctx, pa.minilm_l6_desc(a.compute), synthetic_seed=1)  # same seed: identical replicas
    lo, hi = pa.shard_bounds(a.rows, rank, world)
    s = pa.Searcher(ctx, 384, "cosine")
    s.add_synthetic(1, hi - lo, 0x5EED, first_row=lo, normalize=True)  # MiniLM embeddings are unit-norm
    s.finalize()
    s.set_shard_offset(lo)
    sharded = None
    if world > 1:
        gather = None
        if rehearse:
            def gather(gathered, local):
                out = torch.empty(gathered.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, local.cpu())
                gathered.copy_(out)
        sharded = pa.ShardedSearcher(dist, "cosine", 384, searcher=s, ctx=ctx, device=True, all_gather=gather)
    rng = np.random.default_rng(0)
    ids = rng.integers(1000, 30000, (a.batch, a.seq)).astype(np.int64)
    mask = np.ones_like(ids)
    d0, d1 = a.batch * rank // world, a.batch * (rank + 1) // world  # this rank's documents
    dev = "cpu" if rehearse else "cuda"

    def encode_all():
        emb = m.encode_tokens(ids[d0:d1], mask[d0:d1])
        if world == 1:
            return emb
        mine = torch.from_numpy(np.ascontiguousarray(emb)).to(dev)
        if (d1 - d0) * world == a.batch:
            full = torch.empty((a.batch, emb.shape[1]), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(full, mine)
        else:  # ragged split: gather a list
            parts = [torch.empty((a.batch * (r + 1) // world - a.batch * r // world, emb.shape[1]), dtype=torch.float32, device=dev)
                     for r in range(world)]
            dist.all_gather(parts, mine)
            full = torch.cat(parts)
        return full.cpu().numpy()

    def search_all(emb):
        outs = []
        for q0 in range(0, a.batch, a.pass_queries):
            q = emb[q0:q0 + a.pass_queries]
            outs.append(sharded.search_vectors(None, 10, q) if sharded else s.search_vectors(None, 10, q))
        return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])

    def barrier():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.synchronize()

    enc_ms, scan_ms = [], []
    out = None
    for i in range(a.warmup + a.steps):
        barrier()
        t0 = time.perf_counter()
        emb = encode_all()
        t1 = time.perf_counter()
        out = search_all(emb)
        t2 = time.perf_counter()
        if i >= a.warmup:
            enc_ms.append(1e3 * (t1 - t0))
            scan_ms.append(1e3 * (t2 - t1))
    enc, scan = float(np.mean(enc_ms)), float(np.mean(scan_ms))
    if world > 1:  # slowest rank decides
        t = torch.tensor([enc, scan], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        enc, scan = float(t[0]), float(t[1])
    if rank == 0:
        tot = enc + scan
        print(json.dumps({
            "metric": "end-to-end queries/sec (encode + exact top-10 scan)", "value": a.batch / (tot * 1e-3), "unit": "queries/s",
            "n_gpus": world, "ms_per_step": tot, "encode_ms": enc, "scan_ms": scan,
            "config": {"workload": f"encode batch={a.batch} x seq_len={a.seq} (MiniLM-L6 shape, {a.compute}) + scan {a.rows} x 384, "
                                   f"top-10, {world} MI355X" + (" (encoder data-parallel, corpus row-sharded)" if world > 1 else "")},
            "sample_ids": [int(x) for x in out[0][0][:3]], "sample_scores": [float(x) for x in out[1][0][:3]],
        }), flush=True)
    s.close()
    m.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
