"""The kept profiles cover what the bench line reports (no GPU needed: reads committed files).

bench.py names a dominant kernel per leg; profiles/r03_summary.json (tools/summarize_profiles.py over the rocprofv3 runs of
tools/profile_round.sh) must hold that leg with the profiler's average within the box-to-box spread of the HIP-event figure
bench.py printed in the same process, a roofline fraction that is a fraction, and the PMC traffic of the same command."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _summary():
    with open(os.path.join(ROOT, "profiles", "r03_summary.json")) as f:
        return json.load(f)


def test_every_bench_leg_has_a_kept_profile():
    src = open(os.path.join(ROOT, "bench.py")).read()
    legs = set(re.findall(r'extra\["([a-z0-9_]+)"\]\s*=', src)) | {m for m in re.findall(r'\("off", "([a-z0-9_]+)"\), \("bf16", "([a-z0-9_]+)"\)', src)[0]}
    legs |= {"batch128", "batch256"}  # extra[f"batch{nq}"]
    summary = _summary()
    missing = sorted(l for l in legs if l not in summary)
    assert not missing, missing
    assert "headline" in summary


def test_profiler_and_bench_agree_and_fractions_are_fractions():
    for leg, e in _summary().items():
        if "rocprof_avg_kernel_ms" not in e:
            continue  # encoder / end-to-end legs: per-kernel tables
        assert 0.97 <= e["rocprof_over_bench"] <= 1.03, (leg, e["rocprof_over_bench"])
        assert 0.0 < e["frac_of_8TBps_rocprof"] <= 1.0 and 0.0 < e["frac_of_8TBps_bench"] <= 1.0, leg
        assert e["rocprof_timed_launches"] >= 3, leg
        # PMC pass of the same command: the kernel reads at least what it has to, and not much more
        assert 1.0 <= e["traffic_over_bytes_per_launch"] < 1.25, (leg, e["traffic_over_bytes_per_launch"])
        assert os.path.exists(os.path.join(ROOT, "profiles", "r03_%s_kernel_stats.csv" % ("d768_dot_b64_b128" if leg in ("d768_dot_b64", "d768_dot_b128") else leg))), leg


def test_traffic_table_matches_the_summary():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        traffic = json.load(f)
    summary = _summary()
    for leg, t in traffic.items():
        assert leg in summary and abs(t["bytes_per_launch"] - summary[leg]["traffic_bytes_per_launch"]) < 1.0, leg
        assert t["kernel"] == summary[leg]["kernel"]
