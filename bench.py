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
    ap.add_argument("--screen", default="auto", choices=["auto", "int8", "bf16", "off"],
                    help="resident screening copy of the rows (results do not depend on it): auto = int8; "
                         "off = the scan streams the f32 rows themselves (1536 B/vector)")
    ap.add_argument("--cpu-rows", type=int, default=int(os.environ.get("PCV_BENCH_CPU_ROWS", 1_000_000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective", default=os.environ.get("PCV_BENCH_COLLECTIVE", "torch"), choices=["torch", "native", "both"],
                    help="N>1 hit-list exchange of the TIMED steps: torch.distributed's RCCL group (default), or the library's own RCCL "
                         "communicator (native).  With torch the same steps are run once more through the native communicator AFTER the "
                         "JSON line is out (figures on stderr as a `native_check` line and in gpurun_out/native_check.json): a failure "
                         "there leaves the line as printed; `both` makes such a failure the exit status")
    ap.add_argument("--no-e2e", action="store_true", help="N>1: skip the configs[4] end-to-end leg (encode -> gather -> sharded search)")
    ap.add_argument("--normalized", action="store_true", help="store unit-norm rows (MiniLM-like)")
    ap.add_argument("--clustered", action="store_true",
                    help="main leg on clustered rows (centroid + noise: ~2e4 rows within 0.01 cosine of every query's top-k)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs (config 2, shard, encoder, end to end, clustered)")
    ap.add_argument("--only", default="", help="comma-separated names of the extra legs to run (default: all of them); what "
                    "tools/profile_round.sh uses to put one leg of this file under rocprofv3 at a time")
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


F32_MFMA_PEAK_TFLOPS = 157.3  # exact-f32 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md); the reference computes in f32
CLUSTER_ROWS = 20_000         # rows per cluster of the clustered corpus
CLUSTER_NOISE = 0.004         # cosines inside a cluster spread over ~noise^2 * dim = 0.006: 2e4 rows within 0.01 of a query's top-k


WIPE_GBPS = 30.0  # measured 34 GB/s (tools/scratch notes in DESIGN.md §5): the driver clears freed VRAM in the background


def settle(freed_bytes):
    """After `freed_bytes` of device memory went back to the driver, wait until its background clear of them is over:
    while it runs it takes ~2.6 % of the HBM bandwidth from whatever is being timed."""
    time.sleep(freed_bytes / (WIPE_GBPS * 1e9))


def scan_kernel_name(stats, batch, dim):
    """The dominant kernel of a pass, as rocprofv3 lists it (perceive_amd/csrc/scan_kernels.hip: launch_scan_*)."""
    if stats["kernel_used"] != 2:
        return "scan_wave_kernel"
    if stats["screening_copy"] != 2:
        return "scan_mfma_kernel"
    return "scan_mfma8_hold_kernel" if batch > 64 and (dim + 127) // 128 * 128 <= HOLD_MAX_DIM else "scan_mfma8_kernel"


HOLD_MAX_DIM = 384  # widest rows the block-holding int8 scan takes (csrc/scan_kernels.hip: launch_scan_mfma8)


def roofline_of(st, launches, kernel_ms, rows, dim):
    """Roofline record of one scan leg.  `achieved` = the bytes the selected scan kernel HAS TO move per launch (the
    library's pcv_scan_stats.bytes_streamed: int8 copy + scales, or bf16 copy, or f32 rows + scales) / its hipEvent time;
    `frac` = that / 8 TB/s, <= 1 by construction.  The N*D*4-equivalent rate of the reference formulation (one
    f32 pass, which a scan over a narrow copy does not make) is kept under its own name and is not a fraction of anything."""
    per_launch = st["bytes_streamed"] / max(launches, 1)
    achieved = per_launch / (kernel_ms * 1e-3) / 1e9
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "bytes_per_launch": per_launch, "kernel_ms": kernel_ms,
        "effective_f32_pass_GBps": rows * dim * 4 / (kernel_ms * 1e-3) / 1e9,
    }


def scan_leg(pa, ctx, rows, batch, k, kernel, steps, warmup, seed=0x5EED, clustered=False, searcher=None, dim=384, metric="cosine",
             amplitude=None, tuning=0, keep=False, leg=None):
    """One single-GPU scan measurement: `steps` exact top-k searches of `batch` fresh queries over `rows`
    synthetic rows resident in HBM.  Returns the record that goes under `extra` (same fields as the headline)."""
    own = searcher is None
    if own:
        searcher = pa.Searcher(ctx, dim, metric)
        ncl = max(1, rows // CLUSTER_ROWS) if clustered else 0
        searcher.add_synthetic(1, rows, seed, n_clusters=ncl, noise=CLUSTER_NOISE if clustered else 0.0, amplitude=amplitude)
        searcher.finalize()
    searcher.set_kernel(kernel)
    env_flags = int(os.environ.get("PCV_SCAN_FLAGS", "0"), 0)
    if tuning:
        searcher.set_tuning(env_flags | tuning)
    rng = np.random.default_rng(seed + 17)
    if clustered:
        # queries = unseen members of the corpus' clusters (the oracle's twin generator is test infrastructure;
        # the same construction in numpy: centroid rows come back from the device)
        probe = searcher.get_rows(rng.integers(0, rows, (warmup + steps) * batch))[0]
        q = probe + 0.5 * CLUSTER_NOISE * rng.standard_normal(probe.shape).astype(np.float32)
        queries = q.reshape(warmup + steps, batch, dim).astype(np.float32)
    else:
        queries = rng.standard_normal((warmup + steps, batch, dim)).astype(np.float32)
    searcher.set_mid_copy("auto")
    for i in range(warmup):
        searcher.search_vectors(None, k, queries[i])
    # AUTO decides from the passes it sees whether a mid copy pays and builds it beside the searches (a crowd at the coarse screen
    # two passes in a row; at one query the count hovers around the trigger, so the decision can fall anywhere): the timed passes
    # are one state or the other, not the build in between — a build under way is waited for, and a searcher that has not built
    # one by now keeps none for this leg.
    searcher.wait_background()
    if not searcher.last_stats()["mid_copy"]:
        searcher.search_vectors(None, k, queries[warmup - 1])
        searcher.wait_background()
        if not searcher.last_stats()["mid_copy"]:
            searcher.set_mid_copy("off")
    ctx.synchronize()
    scan_ms, pass_ms, cands, coarse, mids, reruns, launches, spec_reruns, streamed = [], [], [], [], [], 0, 0, 0, 0
    wall = 0.0  # host time inside the search calls (query batch in host memory -> hits in host memory); reading the statistics is not part of a step
    for i in range(steps):
        t0 = time.perf_counter()
        searcher.search_vectors(None, k, queries[warmup + i])
        wall += time.perf_counter() - t0
        st = searcher.last_stats()
        scan_ms.append(st["scan_ms"])
        pass_ms.append(st["total_ms"])
        cands.append(st["candidates"])
        coarse.append(st["coarse_survivors"])
        mids.append(st["mid_survivors"])
        reruns += st["overflow_reruns"]
        spec_reruns += st["speculation_reruns"]
        launches += st["scan_launches"]
        streamed += st["bytes_streamed"]
    st = searcher.last_stats()
    kname = scan_kernel_name(st, batch, dim)
    kernel_ms = float(np.sum(scan_ms)) / max(launches, 1)
    copy = {0: None, 1: "bf16", 2: "int8"}[st["screening_copy"]]
    rec = {
        "workload": f"{rows} x {dim} f32 synthetic" + (" clustered" if clustered else "") +
                    (f" rows x U[{amplitude[0]},{amplitude[1]})" if amplitude else "") + f" corpus, {metric}, batch={batch}, top-{k}, 1 MI355X",
        "kernel": kname, "screening_copy": copy, "ms_per_step": 1e3 * wall / steps, "kernel_ms": kernel_ms,
        "pass_ms": float(np.sum(pass_ms)) / max(launches, 1),
        "fixed_cost_us": 1e3 * (float(np.sum(pass_ms)) - float(np.sum(scan_ms))) / max(launches, 1),
        "roofline": roofline_of({"bytes_streamed": streamed}, launches, kernel_ms, rows, dim),
        "vectors_per_s": rows * steps / wall, "queries_per_s": batch * steps / wall,
        "candidates_per_query": float(np.mean(cands)) / batch, "coarse_survivors_per_query": float(np.mean(coarse)) / batch,
        "mid_copy": bool(st["mid_copy"]), "mid_survivors_per_query": float(np.mean(mids)) / batch,
        "overflow_reruns": reruns, "speculation_reruns": spec_reruns, "steps": steps,
    }
    rec["roofline"]["traffic"], rec["roofline"]["traffic_source"] = measured_traffic(leg, kname, streamed / max(launches, 1))
    rec["roofline"]["traffic_measured_in_this_run"] = False  # (a kept PMC pass of the same command: profiles/)
    if tuning:
        searcher.set_tuning(env_flags)
    if own and not keep:
        searcher.close()
        settle(rows * (dim * 4 + dim + 8))
    return (rec, searcher) if keep else rec


def encoder_leg(pa, ctx, compute, steps=5, warmup=2, batch=256, seq=256, shape="minilm_l6"):
    """BASELINE configs[4]'s encoder (all-MiniLM-L6-v2 shape, 256 documents x 256 tokens), or the shape of the reference's
    default model (MsMarcoBertBaseDotV5 = BERT-base: 12 x 768, 12 heads x 64, FFN 3072; pipeline.rs:76 batches 64 documents);
    seeded synthetic weights."""
    if shape == "minilm_l6":
        desc, label = pa.minilm_l6_desc(compute), "all-MiniLM-L6-v2 shape"
    else:
        desc = pa.make_desc(30522, 768, 12, 12, 3072, 512, pooling="cls", normalize=False, compute=compute)
        label = "BERT-base shape (msmarco-bert-base-dot-v5: 12 x 768, cls pooling, not normalised)"
    m = pa.Model(ctx, desc, synthetic_seed=1)
    rng = np.random.default_rng(0)
    ids = rng.integers(1000, 30000, (batch, seq)).astype(np.int64)
    mask = np.ones((batch, seq), np.int64)
    for _ in range(warmup):
        m.encode_tokens(ids, mask)
    ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        m.encode_tokens(ids, mask)
        ms.append(m.last_stats()["total_ms"])
    wall = time.perf_counter() - t0
    flops = m.last_stats()["flops"]
    m.close()
    dev_ms = float(np.mean(ms))
    tf = flops / (dev_ms * 1e-3) / 1e12
    return {
        "workload": f"encode batch={batch} x seq_len={seq}, {label}, synthetic weights",
        "compute": {"f32": "f32 (exact-f32 MFMA; the reference's dtype)",
                    "bf16x3": "split precision: 3 bf16 terms per f32 operand, 6 bf16 MFMAs per product, f32 accumulate",
                    "f16x2": "split precision: 2 f16 terms per f32 operand, 3 f16 MFMAs per product, f32 accumulate"}[compute],
        "device_ms": dev_ms, "ms_per_step": 1e3 * wall / steps, "tokens_per_s": batch * seq * steps / wall,
        "flops_per_step": flops, "effective_TFLOPps": tf,
        "roofline": {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / F32_MFMA_PEAK_TFLOPS} if compute == "f32" else None,
    }


def e2e_leg(pa, ctx, searcher, rows, steps=5, warmup=4, batch=256, seq=256, k=10):
    """BASELINE configs[4] on one GPU: encode 256 x 256 tokens (f32), then search the 256 embeddings over the
    resident corpus (one pass of 256 queries with the int8 copy at 384-d)."""
    m = pa.Model(ctx, pa.minilm_l6_desc("f32"), synthetic_seed=1)
    rng = np.random.default_rng(1)
    ids = rng.integers(1000, 30000, (batch, seq)).astype(np.int64)
    mask = np.ones((batch, seq), np.int64)
    searcher.set_kernel("auto")
    enc, scan = [], []
    for i in range(warmup + steps):
        ctx.synchronize()
        t0 = time.perf_counter()
        emb = m.encode_tokens(ids, mask)
        t1 = time.perf_counter()
        out = searcher.search_vectors(None, k, emb)
        t2 = time.perf_counter()
        if i >= warmup:
            enc.append(1e3 * (t1 - t0))
            scan.append(1e3 * (t2 - t1))
    m.close()
    e, s_ = float(np.mean(enc)), float(np.mean(scan))
    return {
        "workload": f"encode batch={batch} x seq_len={seq} (MiniLM-L6 shape, f32) + exact top-{k} scan of the {batch} embeddings "
                    f"over {rows} x 384, 1 MI355X",
        "ms_per_step": e + s_, "encode_ms": e, "scan_ms": s_, "queries_per_s": batch / ((e + s_) * 1e-3),
        "sample_ids": [int(x) for x in out[0][0][:3]],
    }


def e2e_dist_leg(pa, ctx, dist, torch, sharded, rank, world, rehearse, rows, barrier, steps=5, warmup=3, batch=256, seq=256, k=10):
    """BASELINE configs[4] on `world` GPUs: data-parallel encode of 256 documents x 256 tokens (f32, all-MiniLM-L6-v2 shape,
    seeded weights), embeddings left on the devices (encode_tokens_device), one all-gather of 256 x 384 floats, then the
    sharded exact top-10 of all 256 over the row-sharded corpus — one pass of 256 queries if every rank keeps the int8 copy of its
    rows, else two of 128 — reading them from device memory (ShardedSearcher.search_device_queries).  Returns rank 0's record (None elsewhere)."""
    m = pa.Model(ctx, pa.minilm_l6_desc("f32"), synthetic_seed=1)
    rng = np.random.default_rng(1)
    ids = rng.integers(1000, 30000, (batch, seq)).astype(np.int64)
    mask = np.ones((batch, seq), np.int64)
    d0, d1 = batch * rank // world, batch * (rank + 1) // world
    D = 384
    emb = torch.zeros((batch, D), dtype=torch.float32, device="cuda")
    even = (d1 - d0) * world == batch

    def one():
        m.encode_tokens_device(ids[d0:d1], mask[d0:d1], emb.data_ptr() + d0 * D * 4)
        mine = emb[d0:d1].clone()
        if rehearse:  # gloo: through the host
            parts = [torch.empty((batch * (r + 1) // world - batch * r // world, D), dtype=torch.float32) for r in range(world)]
            dist.all_gather(parts, mine.cpu())
            emb.copy_(torch.cat(parts))
        elif even:
            dist.all_gather_into_tensor(emb, mine)
        else:
            parts = [torch.empty((batch * (r + 1) // world - batch * r // world, D), dtype=torch.float32, device="cuda") for r in range(world)]
            dist.all_gather(parts, mine)
            emb.copy_(torch.cat(parts))
        torch.cuda.current_stream().synchronize()
        outs = [sharded.search_device_queries(None, k, emb.data_ptr() + q0 * D * 4, min(qpass, batch - q0)) for q0 in range(0, batch, qpass)]
        return np.concatenate([o[0] for o in outs])

    # all 256 queries in one pass if every rank's searcher keeps the int8 copy of all its rows (agreed with one all-reduce: the split
    # of a batch into passes is part of the exchange's protocol, pcv_searcher_allow_wide_sharded_pass); else two passes of 128
    have8 = torch.tensor([1 if sharded.searcher.last_stats()["screening_copy"] == 2 else 0], dtype=torch.int32, device="cpu" if rehearse else "cuda")
    dist.all_reduce(have8, op=dist.ReduceOp.MIN)
    wide = bool(int(have8.item()))
    qpass = batch if wide else 128
    sharded.searcher.allow_wide_sharded_pass(wide)

    for _ in range(warmup):
        got = one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        got = one()
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    m.close()
    sharded.searcher.allow_wide_sharded_pass(False)
    if rank != 0:
        return None
    ms = 1e3 * float(el.item()) / steps
    return {
        "workload": f"encode batch={batch} x seq_len={seq} (MiniLM-L6 shape, f32) data-parallel over {world} GPUs + all-gather of the "
                    f"embeddings (device) + exact top-{k} of the {batch} embeddings over {rows} x 384 row-sharded, {world} MI355X",
        "ms_per_step": ms, "queries_per_s": batch / (ms * 1e-3), "n_gpus": world, "host_hops_of_the_embeddings": 0 if not rehearse else 1,
        "queries_per_pass": qpass,
        "sample_ids": [int(x) for x in got[0][:3]],
    }


def native_check(pa, ctx, dist, torch, searcher, queries, args, rank, world, last, barrier):
    """The timed steps once more through the library's own RCCL communicator (pcv_searcher_search_sharded), after the JSON line
    is out.  Reports on stderr and in gpurun_out/native_check.json; True if it ran and agreed with the torch path."""
    rec = {"native_ok": False, "n_gpus": world}
    comm = None
    try:
        comm = pa.NativeComm.from_dist(ctx, dist)
        k = args.k
        for i in range(args.warmup):
            searcher.search_sharded(comm, None, k, queries[i])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            res = searcher.search_sharded(comm, None, k, queries[args.warmup + i])
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        same = bool((res[0] == last[0]).all()) and bool(np.array_equal(res[1], last[1]))
        # the all-gather entry and the device-query form: this rank's queries gathered on the communicator, searched from device memory
        q = np.ascontiguousarray(queries[args.warmup + args.steps - 1], np.float32)
        per = q.nbytes
        full = torch.zeros(world * q.size, dtype=torch.float32, device="cuda")
        full[rank * q.size:(rank + 1) * q.size] = torch.from_numpy(q.reshape(-1)).cuda()
        torch.cuda.synchronize()
        comm.all_gather(full.data_ptr() + rank * per, full.data_ptr(), per)
        res_dq = searcher.search_sharded_dq(comm, None, k, full.data_ptr(), q.shape[0])  # rank 0's slot: the same queries on every rank
        same_dq = bool((res_dq[0] == last[0]).all())
        rec.update({"native_ok": same and same_dq, "native_ms_per_step": 1e3 * float(el.item()) / args.steps,
                    "native_equals_torch": same, "native_device_queries_equal": same_dq})
    except Exception as e:  # noqa: BLE001 - reported, not raised: the line is out
        rec["native_error"] = f"{type(e).__name__}: {e}"
    finally:
        try:
            if comm is not None:
                comm.close()
        except Exception:
            pass
    if rank == 0:
        print("native_check: " + json.dumps(rec), file=sys.stderr, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "native_check.json"), "w") as f:
                json.dump(rec, f)
        except OSError:
            pass
    return rec["native_ok"]


def measured_traffic(leg, kernel, required_bytes):
    """HBM bytes per launch from the committed PMC passes of this file's legs (profiles/traffic.json, written by
    tools/summarize_profiles.py; FETCH_SIZE x2 + WRITE_SIZE per the gfx950 guide): the entry of THIS leg, and only if
    it was taken on the same kernel with the same required bytes per launch (within 2 %).  bench.py cannot collect PMC
    counters on itself, so this is a figure of the kept profile run of the same command, not of this run; None when
    the leg has no pass, or the pass was of another kernel or size."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(leg)
        if t and t["kernel"].split("<")[0] == kernel and abs(t["required_bytes_per_launch"] / required_bytes - 1.0) < 0.02:
            return t["bytes_per_launch"] * required_bytes / t["required_bytes_per_launch"], t["source"]
    except Exception:
        pass
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
    line_out = sys.stdout
    if use_dist:
        # stdout carries ONE JSON line.  RCCL prints a banner (version, host, library path) on stdout when its first
        # communicator comes up: from here on file descriptor 1 is stderr, and the line goes out through a copy of the real one.
        sys.stdout.flush()
        line_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
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
    # --only with legs that bring their own corpus: the headline shrinks to a token (100k rows through the wave kernel,
    # so that its launches do not share a kernel name with the leg under the profiler) and is marked as skipped
    HEADLINE_LEGS = ("no_guess", "config5_end_to_end", "batch128", "batch256", "f32_rows_b64", "bf16_copy_b64")
    only_legs = [x for x in args.only.split(",") if x]
    token_headline = bool(only_legs) and not any(x in HEADLINE_LEGS for x in only_legs) and world == 1
    if token_headline:
        total_rows, args.kernel = min(total_rows, 100_000), "wave"
    lo = total_rows * rank // world
    hi = total_rows * (rank + 1) // world
    searcher = pa.Searcher(ctx, args.dim, "cosine")
    if args.screen != "auto":
        searcher.set_screening_copy(args.screen)
    t0 = time.time()
    ncl = max(1, total_rows // CLUSTER_ROWS) if args.clustered else 0
    searcher.add_synthetic(1, hi - lo, 0x5EED, first_row=lo, normalize=args.normalized, n_clusters=ncl,
                           noise=CLUSTER_NOISE if args.clustered else 0.0)
    searcher.finalize()
    searcher.set_shard_offset(lo)
    searcher.set_kernel(args.kernel)
    t_build = time.time() - t0

    # same query stream on every rank (device-side generator twin lives in the oracle; queries are
    # tiny, so they are generated with numpy from a fixed seed instead)
    rng = np.random.default_rng(0x5EED + 1)
    queries = rng.standard_normal((args.warmup + args.steps, args.batch, args.dim)).astype(np.float32)
    if args.clustered:  # unseen members of the corpus' clusters: a stored row of this shard + fresh noise
        probe = searcher.get_rows(lo + rng.integers(0, hi - lo, queries.shape[0] * args.batch))[0]
        queries = (probe + 0.5 * CLUSTER_NOISE * queries.reshape(-1, args.dim)).reshape(queries.shape).astype(np.float32)
        if use_dist:  # every rank must search the same queries: rank 0's
            qt = torch.from_numpy(queries).to("cpu" if rehearse else "cuda")
            dist.broadcast(qt, src=0)
            queries = qt.cpu().numpy()

    B, k = args.batch, args.k
    if use_dist:
        # exchange of the [B][k] hit lists: torch.distributed's RCCL group (default), or the library's own
        # persistent RCCL communicator (pcv_comm_*; torch then only bootstraps the id and times the job)
        comm = pa.NativeComm.from_dist(ctx, dist) if args.collective == "native" else None
        want_native_check = args.collective in ("torch", "both") and not rehearse  # (world 1: only under PCV_BENCH_FORCE_DIST, a rehearsal of this code)
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

    scan_ms, pass_ms, host_ms, scan_bytes, streamed_bytes, cands, coarse, reruns, spec_reruns = [], [], [], [], [], [], [], 0, 0
    last = None

    def step(i, timed):
        nonlocal last, reruns, spec_reruns
        q = queries[i]
        if not use_dist:
            last = searcher.search_vectors(None, k, q)
        else:  # local exact top-k -> RCCL all-gather of [B][k] hits -> merge (perceive_amd/sharded.py)
            last = sharded.search_vectors(None, k, q)
        if timed:
            st = searcher.last_stats()
            nl = max(st["scan_launches"], 1)  # > 1 when a pass had to be repeated: the kernel figures are per launch
            pass_ms.append(st["total_ms"] / nl)
            host_ms.append((st["host_enqueue_ms"], st["host_wait_ms"]))
            scan_ms.append(st["scan_ms"] / nl)
            scan_bytes.append(st["bytes_algorithmic"] / nl)
            streamed_bytes.append(st["bytes_streamed"] / nl)
            cands.append(st["candidates"])
            coarse.append(st["coarse_survivors"])
            reruns += st["overflow_reruns"]
            spec_reruns += st["speculation_reruns"]

    for i in range(args.warmup):
        step(i, False)
    # The searcher decides from its first passes whether a mid copy pays (two passes with a crowd at the coarse screen: the third
    # call queues the build, beside the searches): the timed steps are the steady state, so with fewer than three warm-up steps
    # the missing ones are made up for here (untimed, counted in `settle_passes`), and a build under way is waited for.
    settle_passes = max(0, 3 - args.warmup)
    for i in range(settle_passes):
        step(i % max(args.warmup, 1), False)
    searcher.wait_background()
    step(0, False)  # (a pass that would use a copy finished a moment ago; in every rank alike)
    settle_passes += 1
    if not searcher.last_stats()["mid_copy"]:  # (no copy by now: none during the timed steps either — every rank for itself)
        searcher.set_mid_copy("off")
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    barrier()
    elapsed = time.perf_counter() - t0
    searcher.set_mid_copy("auto")

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

    final_out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        vectors_per_s = total_rows * args.steps / elapsed
        st_last = searcher.last_stats()
        copy = {0: None, 1: "bf16", 2: "int8"}[st_last["screening_copy"]]
        per_launch_streamed = float(np.mean(streamed_bytes))  # what this rank's scan kernel has to read per launch
        ids, scores, counts = last
        kname = scan_kernel_name(st_last, B, args.dim)
        roof = roofline_of({"bytes_streamed": per_launch_streamed}, 1, mean_scan_ms, hi - lo, args.dim)
        traffic, traffic_src = measured_traffic("headline" if not args.clustered else "clustered_b64", kname, per_launch_streamed)
        roof.update({
            "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_this_run": False, "kernel": kname, "streams": {None: "f32 rows + row scales", "bf16": "bf16 screening copy",
                                                                                          "int8": "int8 screening copy + one quantisation scale per 32-row block"}[copy],
            "algorithmic_f32_bytes_per_launch": float(np.mean(scan_bytes)),  # N*D*4 of SURVEY 8d: what `effective_f32_pass_GBps` prices
            "kernel_ms_median": float(np.median(scan_ms)), "kernel_ms_min": float(np.min(scan_ms)),
            "read_ceiling_measured": measured_read_ceiling(),  # GB/s, profiles/r01_ubench_hbm_read*.txt
            "pass_ms": float(np.mean(pass_ms)),  # upload + prep_seed + scan + rescore_select on the device
            "fixed_cost_us": 1e3 * (float(np.mean(pass_ms)) - mean_scan_ms),
            "host_enqueue_ms": float(np.mean([h[0] for h in host_ms])), "host_wait_ms": float(np.mean([h[1] for h in host_ms])),
        })
        out = {
            "metric": f"vectors scanned/sec (exact cosine top-{k}, {args.dim}-d f32, batch={B})",
            "value": vectors_per_s,
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_passes": settle_passes,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "screen": ({"int8": "coarse screen = exact integer dot product of int8-quantised rows (resident screening copy, one quantisation scale per "
                                "32-row block: 384 B/vector + 4 B per block) and the int8-quantised query on v_mfma_i32_32x32x32_i8, with a certified "
                                "quantisation margin; ",
                        "bf16": "coarse screen = bf16 MFMA over the resident bf16 screening copy of the rows (768 B/vector); ",
                        None: "rows are read as f32 (1536 B/vector), bf16 MFMA coarse screen; "}[copy]
                       + "the f32 rows are read for the coarse survivors only: exact-f32 fine screen with certified margins, survivors "
                         "ranked in f64: exact top-k") if st_last["kernel_used"] == 2 else
                      "f32 FMA screen with a certified margin, survivors ranked in f64: exact top-k",
            "data": "synthetic clustered" if args.clustered else "synthetic",
            "config": {
                "workload": f"{total_rows} x {args.dim} f32 synthetic corpus, batch={B} queries, top-{k}, "
                            f"{world} MI355X" + (" (rows sharded, RCCL all-gather of per-shard top-k)" if world > 1 else ""),
                "rows": total_rows, "dim": args.dim, "batch": B, "k": k,
                "kernel": {1: "wave", 2: "mfma"}[st_last["kernel_used"]], "screening_copy": copy,
                "rows_normalized": bool(args.normalized), "clustered": bool(args.clustered),
                "collective": (args.collective if use_dist else None),
            },
            "queries_per_s": B * args.steps / elapsed,
            "roofline": roof,
            "candidates_per_query": float(np.mean(cands)) / B,
            "coarse_survivors_per_query": float(np.mean(coarse)) / B,
            "overflow_reruns": reruns,
            "speculation_reruns": spec_reruns,  # passes repeated because a speculative start threshold did not hold (scan.h)
            "build_s": t_build,
            "sample_result": {"ids": [int(x) for x in ids[0][:3]], "scores": [float(x) for x in scores[0][:3]]},
        }
        if world == 1 and not args.no_extra and not use_dist:  # (not in the one-rank rehearsal of the N > 1 path either)
            # the other legs BASELINE.json names, measured in the same run (single GPU only)
            extra = {}
            only = only_legs
            if token_headline:
                out["headline_skipped"] = "--only: the top-level figures are a 100k-row token run, not a measurement"

            def want(*names):
                return not only or any(n in only for n in names)

            es, ew = max(3, min(args.steps, 20)), 6  # (six warm-up passes: AUTO builds the mid copy, where it does, after four)
            # the same workload without the speculative start threshold (scan.h; PCV_SCAN_FLAGS bit 5): the headline's kernel
            # time includes a guess learned from the bench's own i.i.d. queries — this is the figure without it
            if want("no_guess"):
                ng = scan_leg(pa, ctx, total_rows, B, k, args.kernel, max(3, es // 2), ew, searcher=searcher, dim=args.dim, tuning=32)
                out["roofline"]["no_guess"] = {"kernel_ms": ng["kernel_ms"], "frac": ng["roofline"]["frac"], "candidates_per_query": ng["candidates_per_query"]}
                out["roofline"]["no_guess_kernel_ms"] = ng["kernel_ms"]  # (scalars beside the records: a reader that keeps scalars only sees them)
                out["roofline"]["no_guess_frac"] = ng["roofline"]["frac"]
            if args.dim == 384 and want("config5_end_to_end"):
                extra["config5_end_to_end"] = e2e_leg(pa, ctx, searcher, total_rows)
            searcher.set_kernel(args.kernel)
            if copy == "int8":
                # larger batches on the same corpus (the block-holding form of the int8 scan up to 384-d: 128 queries; 256 in one pass)
                for nq in (128, 256):
                    if want(f"batch{nq}"):
                        extra[f"batch{nq}"] = scan_leg(pa, ctx, total_rows, nq, k, args.kernel, max(3, es // 2), ew, searcher=searcher, dim=args.dim, leg=f"batch{nq}")
            if copy is not None and want("f32_rows_b64", "bf16_copy_b64"):
                # the same corpus and queries' shape without the int8 copy: the scan streams the f32 rows themselves
                # (1536 B/vector, the SURVEY 8d / north_star workload) or the bf16 copy; results are the same exact top-k
                held = {"int8": ((args.dim + 127) // 128) * 128 + 0.125, "bf16": 2 * args.dim}  # bytes per row of a copy
                # (a 16-bit mid copy that AUTO built during the 128- / 256-query legs goes with the int8 copy it refines: it is part
                # of what the driver then clears in the background)
                had_mid = bool(searcher.last_stats()["mid_copy"])
                for mode, key in (("off", "f32_rows_b64"), ("bf16", "bf16_copy_b64")):
                    searcher.set_screening_copy(mode)
                    searcher.finalize()
                    if mode == "off":
                        settle(total_rows * (held[copy] + (2 * args.dim + 4 if had_mid else 0)))
                    extra[key] = scan_leg(pa, ctx, total_rows, B, k, args.kernel, es, ew, searcher=searcher, dim=args.dim, leg=key)
                # the north_star's own workload ("coalesced HBM reads of the corpus f32 rows", >= 70 % of the HBM roofline): beside the headline
                f = extra["f32_rows_b64"]
                out["roofline"]["f32_rows"] = {"kernel": f["kernel"], "kernel_ms": f["kernel_ms"], "achieved": f["roofline"]["achieved"],
                                               "frac": f["roofline"]["frac"], "vectors_per_s": f["vectors_per_s"]}
                out["roofline"]["f32_rows_frac"] = f["roofline"]["frac"]
                out["roofline"]["f32_rows_kernel_ms"] = f["kernel_ms"]
                out["roofline"]["f32_rows_achieved"] = f["roofline"]["achieved"]
            if not args.clustered and args.rows >= 1_000_000 and want("clustered_b64", "d768_dot_b64", "d768_dot_b128", "d768_dot_b1"):
                searcher.close()  # two 153.6 GB corpora do not fit: the clustered one replaces the headline one
                searcher = None
                settle(total_rows * (args.dim * 4 + 2 * args.dim + 8))
                # (the searcher's AUTO policy builds the row-major 16-bit mid copy after two passes in a row with more than 4096
                # coarse survivors per query — pcv_searcher_set_mid_copy — and the steady state is what is timed)
                if want("clustered_b64"):
                    extra["clustered_b64"] = scan_leg(pa, ctx, args.rows, B, k, "auto" if token_headline else args.kernel, es, max(ew, 6), clustered=True, dim=args.dim, leg="clustered_b64")
                # the reference's default model (MsMarcoBertBaseDotV5, perceive-cli/state.rs:24): 768-d, dot metric
                # (search.rs:266-279), rows not normalised — norms spread over x[0.5, 2)
                big = max(1_000_000, args.rows // 2)  # as many bytes of rows as the headline corpus
                if want("d768_dot_b64", "d768_dot_b128"):
                    rec, s768 = scan_leg(pa, ctx, big, 64, k, "auto", es, ew, dim=768, metric="dot", amplitude=(0.5, 2.0), keep=True, leg="d768_dot_b64")
                    extra["d768_dot_b64"] = rec
                    extra["d768_dot_b128"] = scan_leg(pa, ctx, big, 128, k, "auto", max(3, es // 2), ew, searcher=s768, dim=768, metric="dot", amplitude=(0.5, 2.0), leg="d768_dot_b128")
                    s768.close()
                    settle(big * (768 * 4 + 768 + 8))
                if want("d768_dot_b1"):
                    extra["d768_dot_b1"] = scan_leg(pa, ctx, max(100_000, args.rows // 10), 1, k, "auto", es, ew, dim=768, metric="dot", amplitude=(0.5, 2.0), leg="d768_dot_b1")
            if want("config2_10m_b1"):
                extra["config2_10m_b1"] = scan_leg(pa, ctx, max(100_000, args.rows // 10), 1, k, "auto", es, ew, leg="config2_10m_b1")
            if want("shard_12p5m_b64"):
                extra["shard_12p5m_b64"] = scan_leg(pa, ctx, max(100_000, args.rows // 8), 64, k, "auto", es, ew, leg="shard_12p5m_b64")
            if want("shard_12p5m_b256"):  # one rank's scan of configs[4]: its 12.5M-row shard against all 256 embeddings in one pass
                extra["shard_12p5m_b256"] = scan_leg(pa, ctx, max(100_000, args.rows // 8), 256, k, "auto", max(3, es // 2), ew, leg="shard_12p5m_b256")
            if want("encoder_256x256"):
                extra["encoder_256x256"] = encoder_leg(pa, ctx, "f32")
            if want("encoder_32x256"):  # one rank's share of configs[4]'s 256 documents (data-parallel encode)
                extra["encoder_32x256"] = encoder_leg(pa, ctx, "f32", batch=32, seq=256, steps=10, warmup=3)
            if want("encoder_64x256"):  # all-MiniLM-L6-v2 at the batch the reference's pipeline forms (sources/pipeline.rs:76)
                extra["encoder_64x256"] = encoder_leg(pa, ctx, "f32", batch=64, seq=256, steps=10, warmup=3)
            if want("encoder_256x256_split_precision"):
                extra["encoder_256x256_split_precision"] = [encoder_leg(pa, ctx, "bf16x3"), encoder_leg(pa, ctx, "f16x2")]
            if want("encoder_bertbase_64x256"):
                extra["encoder_bertbase_64x256"] = encoder_leg(pa, ctx, "f32", batch=64, seq=256, shape="bert_base")
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        final_out = out

    if use_dist and world > 1 and not args.no_e2e and args.dim == 384 and not args.clustered:
        # BASELINE configs[4] at this N: every rank encodes its share of 256 documents, the embeddings stay on the devices,
        # one all-gather completes them on every rank, two sharded passes of 128 queries read them from device memory.  After the
        # headline's figures are final and under a guard: this leg has not run on more than one GPU yet (gloo rehearsal only) —
        # should it hang there, the headline's line still goes out and the job ends
        import threading

        def give_up_e2e():
            if rank == 0:
                final_out["extra"] = {"config5_end_to_end": {"error": "no answer within 240 s"}}
                print(json.dumps(final_out), file=line_out, flush=True)
            os._exit(0)

        guard = threading.Timer(240.0, give_up_e2e)
        guard.daemon = True
        guard.start()
        e2e_dist = e2e_dist_leg(pa, ctx, dist, torch, sharded, rank, world, rehearse, total_rows, barrier)
        guard.cancel()
        if rank == 0:
            final_out["extra"] = {"config5_end_to_end": e2e_dist}
    if rank == 0:
        print(json.dumps(final_out), file=line_out, flush=True)

    native_failed = False
    if use_dist and comm is None and want_native_check and searcher is not None:
        # the library's own communicator (pcv_comm_*, ncclAllGather on its stream, no PyTorch in the data path) on the same
        # steps, after the line above is out: what a Rust / C++ host binds.  It has never had more than one GPU to run on.
        # (a guard: this path has never run on more than one GPU; should it hang there, the line above is out and the job must
        # still end — a timer thread ends the process with the line's status; the main thread may be inside a C call)
        import threading

        def give_up():
            print("native_check: " + json.dumps({"native_ok": False, "n_gpus": world, "native_error": "no answer within 120 s"}), file=sys.stderr, flush=True)
            os._exit(3 if args.collective == "both" else 0)

        guard = threading.Timer(120.0, give_up)
        guard.daemon = True
        guard.start()
        native_failed = not native_check(pa, ctx, dist, torch, searcher, queries, args, rank, world, last, barrier)
        guard.cancel()
    if use_dist and comm is not None:
        comm.close()
    if searcher is not None:
        searcher.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if native_failed and args.collective == "both":
        raise SystemExit(3)


if __name__ == "__main__":
    main()
