#!/usr/bin/env python3
"""Print one bench.py JSON line (file argument) as a table: python tools/show_bench.py gpurun_out/bench.log"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f'headline {d["config"]["workload"]}: {d["value"]/1e9:.2f} G vectors/s  step {d["ms_per_step"]:.3f} ms  kernel {r["kernel"]} {r["kernel_ms"]:.3f} ms  '
      f'{r["achieved"]:.0f} GB/s frac {r["frac"]:.3f}  traffic {r.get("traffic")}  fixed {r["fixed_cost_us"]:.0f} us')
print("  no_guess", r.get("no_guess"), " f32_rows", r.get("f32_rows"))
print(f'  cand/q {d["candidates_per_query"]:.0f} coarse/q {d.get("coarse_survivors_per_query", 0):.0f} reruns {d["overflow_reruns"]},{d["speculation_reruns"]}')
for k, v in d.get("extra", {}).items():
    if isinstance(v, list):
        for x in v:
            print(f'{k:24s} {x["compute"][:28]:28s} {x["device_ms"]:.3f} ms  {x["effective_TFLOPps"]:.1f} TF')
    elif isinstance(v.get("roofline"), dict) and v["roofline"].get("bound") == "hbm":
        print(f'{k:24s} {v["kernel"]:24s} kernel {v["kernel_ms"]:.3f}  step {v["ms_per_step"]:.3f}  frac {v["roofline"]["frac"]:.3f}  {v["roofline"]["achieved"]:.0f} GB/s  '
              f'cand/q {v["candidates_per_query"]:.0f} coarse/q {v["coarse_survivors_per_query"]:.0f} reruns {v["overflow_reruns"]},{v["speculation_reruns"]} copy {v["screening_copy"]}')
    elif isinstance(v.get("roofline"), dict):
        print(f'{k:24s} {v["device_ms"]:.3f} ms  {v["effective_TFLOPps"]:.1f} TF  frac {v["roofline"]["frac"]:.3f}')
    else:
        print(f'{k:24s}', {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "workload"})
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
