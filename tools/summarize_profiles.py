#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one round (gpurun_out/...) into the committed summaries under
profiles/: kernel-stats CSVs are copied as they are; PMC passes are reduced to per-launch HBM
traffic, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950
(FETCH_SIZE counts 64 B per 128 B request on wide coalesced streaming reads: x2; WRITE_SIZE exact;
both in KiB).

    python tools/summarize_profiles.py r01
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


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep only the latest run of a directory.  `pattern` is
    <dir>/*/<file>; the file may also sit directly in <dir>."""
    flat = pattern.replace(os.sep + "*" + os.sep, os.sep)
    files = sorted(set(glob.glob(pattern)) | set(glob.glob(flat)), key=os.path.getmtime)
    return files[-1:]


def pmc_mean(dirname, counter, kernel_substr):
    vals = []
    for f in newest(os.path.join(GO, dirname, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def kernel_avg_ms(dirname, kernel_substr):
    for f in newest(os.path.join(GO, dirname, "*", "*_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Name"]:
                return float(row["AverageNs"]) / 1e6, int(row["Calls"])
    return None, 0


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(OUT, exist_ok=True)
    summary = {}
    cases = [
        # (label, stats dir, fetch dir, write dir, kernel, rows, dim)
        ("100m_b64_int8", "prof_default", "pmc_fetch", "pmc_write", "scan_mfma8_kernel", 100_000_000, 384),
        ("100m_b64_bf16", "prof_default_bf16", "pmc_fetch_bf16", "pmc_write_bf16", "scan_mfma_kernel", 100_000_000, 384),
        ("100m_b64_f32rows", "prof_default_f32", None, None, "scan_mfma_kernel", 100_000_000, 384),
        ("10m_b1_int8", "prof_10m_b1", "pmc_fetch_10m_b1", "pmc_write_10m_b1", "scan_mfma8_kernel", 10_000_000, 384),
        ("12p5m_b64_int8", "prof_12p5m", None, None, "scan_mfma8_kernel", 12_500_000, 384),
        ("100m_b64_int8_clustered", "prof_clustered", None, None, "scan_mfma8_kernel", 100_000_000, 384),
    ]
    for c in ("f32", "bf16x3", "f16x2"):  # encoder forward per kernel
        for f in newest(os.path.join(GO, f"prof_enc_{c}", "*", "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(OUT, f"{tag}_encode_b256_l256_{c}_kernel_stats.csv"))
    for label, sdir, fdir, wdir, kern, rows, dim in cases:
        for f in newest(os.path.join(GO, sdir, "*", "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(OUT, f"{tag}_{label}_kernel_stats.csv"))
        ms, calls = kernel_avg_ms(sdir, kern)
        fetch_kib, nf = pmc_mean(fdir, "FETCH_SIZE", kern) if fdir else (None, 0)
        write_kib, nw = pmc_mean(wdir, "WRITE_SIZE", kern) if wdir else (None, 0)
        alg = rows * dim * 4
        fixed = {}
        for f in newest(os.path.join(GO, sdir, "*", "*_kernel_stats.csv")):
            for row in csv.DictReader(open(f)):
                for k in ("upload_kernel", "prep_seed", "rescore_select", "quantize_queries"):
                    if k in row["Name"]:
                        fixed[k + "_avg_us"] = float(row["AverageNs"]) / 1e3
        entry = {
            "kernel": kern, "rows": rows, "dim": dim, "algorithmic_bytes_per_launch": alg, "other_kernels_of_a_pass": fixed,
            "rocprof_avg_kernel_ms": ms, "rocprof_calls": calls,
            "achieved_GBps_from_rocprof": alg / (ms * 1e-3) / 1e9 if ms else None,
            "FETCH_SIZE_KiB_raw": fetch_kib, "FETCH_launches": nf,
            "WRITE_SIZE_KiB_raw": write_kib, "WRITE_launches": nw,
        }
        if fetch_kib is not None:
            read_b = fetch_kib * 1024 * 2.0  # gfx950 correction for 16 B/lane streaming reads
            write_b = (write_kib or 0.0) * 1024
            entry["hbm_read_bytes_per_launch"] = read_b
            entry["hbm_write_bytes_per_launch"] = write_b
            entry["traffic_bytes_per_launch"] = read_b + write_b
            entry["traffic_over_algorithmic"] = (read_b + write_b) / alg
            entry["traffic_bytes_per_row"] = (read_b + write_b) / rows
        summary[label] = entry
    with open(os.path.join(OUT, f"{tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # what bench.py reads to fill roofline.traffic (per-row figure of the newest round)
    with open(os.path.join(OUT, "traffic.json"), "w") as f:
        table = {}
        for k, e in summary.items():  # the first (largest) measured configuration of a kernel wins
            if e.get("traffic_bytes_per_row") and e["kernel"] not in table:
                table[e["kernel"]] = {"bytes_per_row": e["traffic_bytes_per_row"], "dim": e["dim"],
                                      "source": f"profiles/{tag}_summary.json:{k}"}
        json.dump(table, f, indent=1)
    # encoder forward: per-kernel time per forward and TFLOP/s of the GEMM shapes (MiniLM-L6 shape, 256 x 256 tokens)
    enc = {}
    for c in ("f32", "bf16x3", "f16x2"):
        for f in newest(os.path.join(GO, f"prof_enc_{c}", "*", "*_kernel_stats.csv")):
            rows_ = list(csv.DictReader(open(f)))
            fwd = 9  # bench_encode.py --steps 7 --warmup 2
            tot = sum(float(r["TotalDurationNs"]) for r in rows_ if "synth_weights" not in r["Name"] and "split_planes" not in r["Name"])
            enc[c] = {"ms_per_forward": tot / 1e6 / fwd, "effective_TFLOPps": 1.546188e12 / (tot / 1e9 / fwd) / 1e12,
                      "kernels_us_per_forward": {r["Name"].split("(anonymous namespace)::")[-1].split("(")[0]:
                                                 round(float(r["TotalDurationNs"]) / 1e3 / fwd, 1) for r in rows_
                                                 if float(r["TotalDurationNs"]) / 1e3 / fwd > 5}}
    if enc:
        summary["encoder_256x256"] = enc
        with open(os.path.join(OUT, f"{tag}_summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
