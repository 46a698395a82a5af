"""The kept profiles cover what the bench line reports (no GPU needed: reads committed files).

bench.py names a dominant kernel per leg; profiles/r03_summary.json (tools/summarize_profiles.py over the rocprofv3 runs of
tools/profile_round.sh) must hold that leg with the profiler's average within the box-to-box spread of the HIP-event figure
bench.py printed in the same process, a roofline fraction that is a fraction, and the PMC traffic of the same command."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _round():
    """The newest round that has a summary of its profile runs under profiles/."""
    import glob
    return max(int(re.search(r"r(\d+)_summary", f).group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))


def _summary():
    with open(os.path.join(ROOT, "profiles", "r%02d_summary.json" % _round())) as f:
        return json.load(f)


ADDED_IN = {"shard_12p5m_b256": 4, "encoder_32x256": 4, "encoder_64x256": 4}  # legs newer than round 3: required from that round's summary on


def test_every_bench_leg_has_a_kept_profile():
    src = open(os.path.join(ROOT, "bench.py")).read()
    legs = set(re.findall(r'extra\["([a-z0-9_]+)"\]\s*=', src)) | {m for m in re.findall(r'\("off", "([a-z0-9_]+)"\), \("bf16", "([a-z0-9_]+)"\)', src)[0]}
    legs |= {"batch128", "batch256"}  # extra[f"batch{nq}"]
    summary = _summary()
    missing = sorted(l for l in legs if l not in summary and ADDED_IN.get(l, 0) <= _round())
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
        # (the x2 of the gfx950 correction is right for the 16-byte-per-lane row stream and doubles what the survivors' narrow reads
        # — mid rows, query rows — really moved: the clustered corpus, 20 000 coarse survivors per query, sits at 1.2 by that count)
        assert 1.0 <= e["traffic_over_bytes_per_launch"] < 1.35, (leg, e["traffic_over_bytes_per_launch"])
        assert os.path.exists(os.path.join(ROOT, "profiles", "r%02d_%s_kernel_stats.csv" % (_round(), "d768_dot_b64_b128" if leg in ("d768_dot_b64", "d768_dot_b128") else leg))), leg


def test_traffic_table_matches_the_summary():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        traffic = json.load(f)
    summary = _summary()
    for leg, t in traffic.items():
        assert leg in summary and abs(t["bytes_per_launch"] - summary[leg]["traffic_bytes_per_launch"]) < 1.0, leg
        assert t["kernel"] == summary[leg]["kernel"]
        assert t["source"].endswith(":" + leg), (leg, t["source"])  # bench.py looks a leg's traffic up under the leg's own name


def test_bench_looks_traffic_up_by_leg_name():
    # bench.py: measured_traffic(leg, kernel, bytes) returns the entry of that leg or nothing (round 3 returned the first entry
    # of the table whose kernel and size matched: the 768-d legs carried the headline's PMC bytes)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        traffic = json.load(f)
    for leg, t in traffic.items():
        got, src = bench.measured_traffic(leg, t["kernel"].split("<")[0], t["required_bytes_per_launch"])
        assert src == t["source"] and abs(got - t["bytes_per_launch"]) < 1.0, leg
        assert bench.measured_traffic(leg, "some_other_kernel", t["required_bytes_per_launch"]) == (None, None)
        assert bench.measured_traffic(leg, t["kernel"].split("<")[0], 2.0 * t["required_bytes_per_launch"]) == (None, None)
    assert bench.measured_traffic("no_such_leg", "scan_mfma8_kernel", 1.0) == (None, None)
