#!/usr/bin/env python3
"""bench.py — the scan benchmark BASELINE.json names.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--batch B]

A step = one exact top-10 search of a batch of queries over the whole synthetic corpus (the hot
path of perceive-core: lib.rs:63-77 similarity + search.rs:157-182 selection), corpus resident in
HBM.  N=1 default workload: BASELINE configs[2] "100M x 384-d, batch=64 queries, top-10, 1 MI355X"
(the configuration the >=70 %-of-HBM-roofline target is quoted on; it fits one 288 GB GPU).
N>1 (launched by torch.distributed.run, one rank per GPU): the same 100M rows row-sharded over the
ranks, per-shard exact top-k, RCCL all-gather of the [B][k] hit lists, merge — strong scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=int(os.environ.get("PCV_BENCH_ROWS", 100_000_000)))
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PCV_BENCH_BATCH", 64)))
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--kernel", default="auto", choices=["auto", "wave", "mfma"])
    ap.add_argument("--cpu-rows", type=int, default=int(os.environ.get("PCV_BENCH_CPU_ROWS", 1_000_000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective", default=os.environ.get("PCV_BENCH_COLLECTIVE", "torch"), choices=["torch", "native"],
                    help="N>1 hit-list exchange: torch.distributed's RCCL group, or the library's own RCCL communicator")
    ap.add_argument("--normalized", action="store_true", help="store unit-norm rows (MiniLM-like)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = --rows is the whole corpus, sharded (BASELINE configs[3]); weak = --rows per GPU")
    return ap.parse_args()


def effective_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands
    one GPU's share of the host — 16 of 256 hardware threads — to the job; starting 256 threads under a
    16-CPU quota only adds throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(args):
    """Reference-shaped CPU leg: the oracle's multithreaded C port of lib.rs:67-77 + top-k on the
    host cores of this node, bounded sample (reported, not a target)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi

    orc = oracle_ffi.load()
    threads = min(orc.hardware_threads(), effective_cpus())
    n = max(10_000, args.cpu_rows)
    rows = orc.synth_rows(0x5EED, 0, n, args.dim, args.normalized)
    q = orc.synth_rows(0x5EED + 1, 0, args.batch, args.dim)
    orc.baseline_scan(q, rows[: max(10_000, n // 10)], args.k, threads)  # warm the thread pool / caches
    total, reps = 0.0, 0
    while total < 12.0 and reps < 2000:  # ~12 s of CPU work in all
        secs, _, _ = orc.baseline_scan(q, rows, args.k, threads)
        total += secs
        reps += 1
    # the reference-shaped form (normalised copy of the corpus, full [B,N] score matrix, then select:
    # what libtorch does for lib.rs:73-77) on a slice small enough for its 4*B*N-byte score matrix
    ns = min(n, 250_000)
    shaped_s, _, _ = orc.baseline_scan(q, rows[:ns], args.k, threads, shaped=True)
    return {
        "value": n * reps / total,
        "unit": "vectors/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{reps} x fused f32 cosine+top-{args.k} scan of {n} x {args.dim} synthetic rows, batch={args.batch}, "
                  f"{total:.1f} s on {threads} threads = the job's CPU quota of {orc.hardware_threads()} hardware threads (oracle/baseline.c)",
        "queries_per_s": args.batch * reps / total,
        "reference_shaped_vectors_per_s": ns / shaped_s,
    }


def measured_traffic(kernel, rows, dim):
    """HBM bytes per launch from the committed PMC pass of this command (profiles/traffic.json,
    written by tools/summarize_profiles.py; FETCH_SIZE x2 + WRITE_SIZE per the gfx950 guide), scaled by
    rows.  bench.py cannot collect PMC counters on itself; None when no pass covers this kernel."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel]
        if t["dim"] != dim or not t["bytes_per_row"]:
            return None, None
        return t["bytes_per_row"] * rows, t["source"]
    except Exception:
        return None, None


def measured_read_ceiling():
    """Best streaming-read rate of tools/ubench/hbm_read.hip / hbm_read2.hip on this part (committed outputs
    of the round's runs; plain load-only kernels over the same 153.6 GB, several address patterns).  Extra context for `roofline`; `peak` stays the
    8 TB/s of the microarchitecture guide."""
    best = 0.0
    try:
        for line in open(os.path.join(ROOT, "profiles", "r01_ubench_hbm_read.txt")):
            if "GB/s best" in line:
                best = max(best, float(line.split("GB/s avg,")[1].split("GB/s best")[0]))
        for line in open(os.path.join(ROOT, "profiles", "r01_ubench_hbm_read2.txt")):  # address-pattern variants
            if line.startswith("pattern") and line.rstrip().endswith("GB/s"):
                best = max(best, float(line.split()[-2]))
    except Exception:
        pass
    return best or None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = torch = None
    use_dist = world > 1 or os.environ.get("PCV_BENCH_FORCE_DIST") == "1"  # the latter: 1-rank rehearsal
    if use_dist:
        # torch first: its bundled HIP runtime must be the one both it and libperceive_hip.so bind
        import torch
        import torch.distributed as dist

        # PCV_BENCH_REHEARSE=1: every rank on GPU 0 over gloo — exercises the N>1 code path (sharding,
        # exchange, merge, reporting) on a one-GPU box; the numbers it prints mean nothing
        rehearse = os.environ.get("PCV_BENCH_REHEARSE") == "1"
        if rehearse:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import perceive_amd as pa

    ctx = pa.Context(local_rank if use_dist else 0)
    total_rows = args.rows * world if args.scaling == "weak" else args.rows
    lo = total_rows * rank // world
    hi = total_rows * (rank + 1) // world
    searcher = pa.Searcher(ctx, args.dim, "cosine")
    t0 = time.time()
    searcher.add_synthetic(1, hi - lo, 0x5EED, first_row=lo, normalize=args.normalized)
    searcher.finalize()
    searcher.set_shard_offset(lo)
    searcher.set_kernel(args.kernel)
    t_build = time.time() - t0

    # same query stream on every rank (device-side generator twin lives in the oracle; queries are
    # tiny, so they are generated with numpy from a fixed seed instead)
    rng = np.random.default_rng(0x5EED + 1)
    queries = rng.standard_normal((args.warmup + args.steps, args.batch, args.dim)).astype(np.float32)

    B, k = args.batch, args.k
    if use_dist:
        # exchange of the [B][k] hit lists: torch.distributed's RCCL group (default), or the library's own
        # persistent RCCL communicator (pcv_comm_*; torch then only bootstraps the id and times the job)
        comm = pa.NativeComm.from_dist(ctx, dist) if args.collective == "native" else None
        gather = None
        if rehearse:  # gloo has no device all-gather: stage through the host (torch copies on the current stream)
            def gather(gathered, local):
                h = local.cpu()
                out = torch.empty(gathered.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, h)
                gathered.copy_(out)
        sharded = pa.ShardedSearcher(dist, "cosine", args.dim, searcher=searcher, ctx=ctx, device=True, comm=comm,
                                     all_gather=gather)

    def barrier():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.synchronize()

    scan_ms, pass_ms, host_ms, scan_bytes, cands, reruns = [], [], [], [], [], 0
    last = None

    def step(i, timed):
        nonlocal last, reruns
        q = queries[i]
        if not use_dist:
            last = searcher.search_vectors(None, k, q)
        else:  # local exact top-k -> RCCL all-gather of [B][k] hits -> merge (perceive_amd/sharded.py)
            last = sharded.search_vectors(None, k, q)
        if timed:
            st = searcher.last_stats()
            pass_ms.append(st["total_ms"])
            host_ms.append((st["host_enqueue_ms"], st["host_wait_ms"]))
            scan_ms.append(st["scan_ms"])
            scan_bytes.append(st["bytes_algorithmic"])
            cands.append(st["candidates"])
            reruns += st["overflow_reruns"]

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    barrier()
    elapsed = time.perf_counter() - t0

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the dominant kernel's duration: slowest rank per step decides
        sm = torch.tensor([float(np.mean(scan_ms))], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(sm, op=dist.ReduceOp.MAX)
        mean_scan_ms = float(sm.item())
    else:
        mean_scan_ms = float(np.mean(scan_ms))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        vectors_per_s = total_rows * args.steps / elapsed
        per_launch_bytes = float(np.mean(scan_bytes))  # this rank's shard: rows * dim * 4
        achieved = per_launch_bytes / (mean_scan_ms * 1e-3) / 1e9
        ids, scores, counts = last
        kname = "scan_mfma_kernel" if searcher.last_stats()["kernel_used"] == 2 else "scan_wave_kernel"
        traffic, traffic_src = measured_traffic(kname, (hi - lo), args.dim)
        out = {
            "metric": f"vectors scanned/sec (exact cosine top-{k}, {args.dim}-d f32, batch={B})",
            "value": vectors_per_s,
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{total_rows} x {args.dim} f32 synthetic corpus, batch={B} queries, top-{k}, "
                            f"{world} MI355X" + (" (rows sharded, RCCL all-gather of per-shard top-k)" if world > 1 else ""),
                "rows": total_rows, "dim": args.dim, "batch": B, "k": k,
                "kernel": {1: "wave", 2: "mfma"}[searcher.last_stats()["kernel_used"]],
                "rows_normalized": bool(args.normalized),
                "collective": (args.collective if use_dist else None),
            },
            "queries_per_s": B * args.steps / elapsed,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": kname,
                "bytes_per_launch": per_launch_bytes,
                "kernel_ms": mean_scan_ms,
                "kernel_ms_median": float(np.median(scan_ms)),
                "kernel_ms_min": float(np.min(scan_ms)),
                "read_ceiling_measured": measured_read_ceiling(),  # GB/s, profiles/r01_ubench_hbm_read*.txt
                "pass_ms": float(np.mean(pass_ms)),  # prep + seed + scan + rescore + select on the device
                "host_enqueue_ms": float(np.mean([h[0] for h in host_ms])),
                "host_wait_ms": float(np.mean([h[1] for h in host_ms])),
            },
            "candidates_per_query": float(np.mean(cands)) / B,
            "overflow_reruns": reruns,
            "build_s": t_build,
            "sample_result": {"ids": [int(x) for x in ids[0][:3]], "scores": [float(x) for x in scores[0][:3]]},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)

    if use_dist and comm is not None:
        comm.close()
    searcher.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
