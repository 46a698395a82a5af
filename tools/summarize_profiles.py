#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one round (gpurun_out/prof_<leg>, pmc_fetch_<leg>, pmc_write_<leg>, written by
tools/profile_round.sh) into the committed summaries under profiles/:

  <tag>_<leg>_kernel_stats.csv   the `--kernel-trace --stats` table of the leg's bench.py command, as rocprofv3 wrote it
  <tag>_summary.json             per leg: the kernel bench.py names, its average duration under the profiler and from
                                 bench.py's own HIP events in the same process, the bytes it has to move per launch,
                                 the roofline fraction from either clock, and the PMC traffic per launch
  traffic.json                   per leg HBM bytes per launch; bench.py reads it for `roofline.traffic`

PMC passes are corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950 (FETCH_SIZE counts 64 B
per 128 B request on wide coalesced streaming reads: x2; WRITE_SIZE exact; both in KiB).

    python tools/summarize_profiles.py r04
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
GO = os.path.join(ROOT, "gpurun_out")
PEAK = 8000.0  # GB/s

# leg of bench.py -> (profile directory suffix, where its record sits in the JSON line the run printed)
SCAN_LEGS = [
    ("headline", "headline", None),
    ("f32_rows_b64", "f32_rows_b64", None),
    ("bf16_copy_b64", "bf16_copy_b64", None),
    ("batch128", "batch128", "batch128"),
    ("batch256", "batch256", "batch256"),
    ("clustered_b64", "clustered_b64", None),
    ("d768_dot_b64", "d768_dot_b64_b128", "d768_dot_b64"),
    ("d768_dot_b128", "d768_dot_b64_b128", "d768_dot_b128"),
    ("d768_dot_b1", "d768_dot_b1", "d768_dot_b1"),
    ("config2_10m_b1", "config2_10m_b1", None),
    ("shard_12p5m_b64", "shard_12p5m_b64", None),
    ("shard_12p5m_b256", "shard_12p5m_b256", "shard_12p5m_b256"),
]
ENC_LEGS = ["encoder_256x256", "encoder_256x256_split_precision", "encoder_bertbase_64x256", "encoder_32x256", "encoder_64x256", "config5_end_to_end"]


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep only the latest run of a directory.  `pattern` is
    <dir>/*/<file>; the file may also sit directly in <dir>."""
    flat = pattern.replace(os.sep + "*" + os.sep, os.sep)
    files = sorted(set(glob.glob(pattern)) | set(glob.glob(flat)), key=os.path.getmtime)
    return files[-1:]


def bench_line(logname):
    """The JSON line bench.py printed inside a profiler run's log."""
    try:
        for line in reversed(open(os.path.join(GO, logname + ".log")).read().splitlines()):
            if line.startswith('{"metric"'):
                return json.loads(line)
    except OSError:
        pass
    return None


def stats_rows(dirname):
    for f in newest(os.path.join(GO, dirname, "*", "*_kernel_stats.csv")):
        return f, list(csv.DictReader(open(f)))
    return None, []


def pick_kernel(rows, base, kernel_ms):
    """Of the kernels of a run whose name contains `base`, the one whose average is nearest to the duration bench.py
    measured for the leg (a run holds at most the token headline's wave kernel besides the leg's own)."""
    cand = [r for r in rows if base + "<" in r["Name"] or base + "(" in r["Name"]]
    if not cand:
        return None
    return min(cand, key=lambda r: abs(float(r["AverageNs"]) / 1e6 - kernel_ms))


def timed_avg_ms(dirname, kernel_name, n):
    """Average duration of the last n dispatches of a kernel in a run (the timed steps of the leg; --stats also counts the
    warm-up passes, which run before a guess has been learned and, on the clustered corpus, before the mid copy exists)."""
    for f in newest(os.path.join(GO, dirname, "*", "durations.json")):
        d = json.load(open(f)).get(kernel_name)
        if d:
            d = d[-n:]
            return sum(d) / len(d) / 1e6, len(d)
    return None, 0


def pmc_mean(dirname, counter, kernel_name):
    vals = []
    for f in newest(os.path.join(GO, dirname, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and row["Kernel_Name"] == kernel_name:
                vals.append(float(row["Counter_Value"]))
    if len(vals) > 2:
        vals = vals[1:]  # the first launch of a process also pages the code object in
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def short(name):
    """`void pcv::(anonymous namespace)::scan_mfma8_kernel<2, true, 4, 3>(pcv::ScanParams const*)` -> `scan_mfma8_kernel<2, true, 4, 3>`"""
    n = name.replace("void ", "").replace("pcv::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in n:  # cut at the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace("pcv::", "")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    os.makedirs(OUT, exist_ok=True)
    summary, traffic = {}, {}
    for leg, suffix, key in SCAN_LEGS:
        out = bench_line("prof_" + suffix)
        f, rows = stats_rows("prof_" + suffix)
        if out is None or f is None:
            print(f"(no profile of {leg})")
            continue
        shutil.copy(f, os.path.join(OUT, f"{tag}_{suffix}_kernel_stats.csv"))
        if key is None:
            rec = {"kernel": out["roofline"]["kernel"], "kernel_ms": out["roofline"]["kernel_ms"],
                   "bytes_per_launch": out["roofline"]["bytes_per_launch"], "workload": out["config"]["workload"],
                   "ms_per_step": out["ms_per_step"], "candidates_per_query": out["candidates_per_query"], "steps": out["steps"]}
        else:
            e = out["extra"][key]
            rec = {"kernel": e["kernel"], "kernel_ms": e["kernel_ms"], "bytes_per_launch": e["roofline"]["bytes_per_launch"],
                   "workload": e["workload"], "ms_per_step": e["ms_per_step"], "candidates_per_query": e["candidates_per_query"],
                   "steps": e["steps"] + e["overflow_reruns"] + e["speculation_reruns"]}
        row = pick_kernel(rows, rec["kernel"], rec["kernel_ms"])
        if row is None:
            print(f"(kernel {rec['kernel']} not in {f})")
            continue
        all_ms = float(row["AverageNs"]) / 1e6
        prof_ms, n_timed = timed_avg_ms("prof_" + suffix, row["Name"], rec["steps"])
        if prof_ms is None:
            prof_ms, n_timed = all_ms, int(row["Calls"])
        entry = {
            "workload": rec["workload"], "command": "bench.py --no-cpu-baseline " + ("--only " + key if key else "--no-extra") + " (tools/profile_round.sh)",
            "kernel": short(row["Name"]), "bytes_per_launch": rec["bytes_per_launch"],
            "rocprof_avg_kernel_ms": prof_ms, "rocprof_timed_launches": n_timed,
            "rocprof_avg_kernel_ms_all_launches": all_ms, "rocprof_calls": int(row["Calls"]),
            "bench_hip_event_kernel_ms": rec["kernel_ms"], "rocprof_over_bench": prof_ms / rec["kernel_ms"],
            "frac_of_8TBps_rocprof": rec["bytes_per_launch"] / (prof_ms * 1e-3) / 1e9 / PEAK,
            "frac_of_8TBps_bench": rec["bytes_per_launch"] / (rec["kernel_ms"] * 1e-3) / 1e9 / PEAK,
            "ms_per_step_under_profiler": rec["ms_per_step"], "candidates_per_query": rec["candidates_per_query"],
            "other_kernels_of_a_pass_avg_us": {short(r["Name"]): round(float(r["AverageNs"]) / 1e3, 1) for r in rows
                                               if any(k in r["Name"] for k in ("upload_kernel", "prep_seed", "rescore_select", "quantize_queries"))},
        }
        # PMC passes of the same command (fewer steps): counters of exactly this kernel name
        pout = bench_line("pmc_fetch_" + suffix)
        fetch_kib, nf = pmc_mean("pmc_fetch_" + suffix, "FETCH_SIZE", row["Name"])
        write_kib, nw = pmc_mean("pmc_write_" + suffix, "WRITE_SIZE", row["Name"])
        if fetch_kib is not None:
            read_b = fetch_kib * 1024 * 2.0  # gfx950 correction for 16 B/lane streaming reads
            write_b = (write_kib or 0.0) * 1024
            entry.update({"FETCH_SIZE_KiB_raw": fetch_kib, "FETCH_launches": nf, "WRITE_SIZE_KiB_raw": write_kib, "WRITE_launches": nw,
                          "hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b,
                          "traffic_bytes_per_launch": read_b + write_b,
                          "traffic_over_bytes_per_launch": (read_b + write_b) / rec["bytes_per_launch"]})
            traffic[leg] = {"bytes_per_launch": read_b + write_b, "required_bytes_per_launch": rec["bytes_per_launch"],
                            "kernel": entry["kernel"], "clustered": "clustered" in leg, "source": f"profiles/{tag}_summary.json:{leg}"}
        elif pout is None:
            entry["pmc"] = None
        summary[leg] = entry
    # encoder legs: per-kernel time per forward
    for leg in ENC_LEGS:
        out = bench_line("prof_" + leg)
        f, rows = stats_rows("prof_" + leg)
        if out is None or f is None:
            print(f"(no profile of {leg})")
            continue
        shutil.copy(f, os.path.join(OUT, f"{tag}_{leg}_kernel_stats.csv"))
        e = out["extra"][leg]
        es = e if isinstance(e, list) else [e]
        kern = {short(r["Name"]): {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3)}
                for r in rows if any(k in r["Name"] for k in ("gemm", "attention", "embed", "pool", "layer_norm", "scan_mfma8", "rescore", "prep_seed", "quantize"))}
        summary[leg] = {"bench": [{k: x.get(k) for k in ("workload", "compute", "device_ms", "ms_per_step", "encode_ms", "scan_ms", "effective_TFLOPps", "roofline") if k in x} for x in es],
                        "kernels_under_profiler": kern}
    with open(os.path.join(OUT, f"{tag}_summary.json"), "w") as fo:
        json.dump(summary, fo, indent=1)
    if traffic:
        with open(os.path.join(OUT, "traffic.json"), "w") as fo:
            json.dump(traffic, fo, indent=1)
    for leg, e in summary.items():
        if "rocprof_avg_kernel_ms" in e:
            print(f"{leg:20s} {e['kernel'][:44]:44s} rocprof {e['rocprof_avg_kernel_ms']:.3f} ms  bench {e['bench_hip_event_kernel_ms']:.3f} ms  "
                  f"ratio {e['rocprof_over_bench']:.3f}  frac {e['frac_of_8TBps_rocprof']:.3f}  traffic/required {e.get('traffic_over_bytes_per_launch')}")


if __name__ == "__main__":
    main()
