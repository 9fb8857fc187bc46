"""CPU: the ONE stdout line of bench.py stays parseable by the driver.  Round 5's line grew to 24 958 bytes (ten full sub-workload
objects) and the driver's ~20 000-byte stdout capture cut it mid-token: `BENCH_r05.json.parsed` was null and the round counted as
unmeasured.  bench.compact_line is a pure function of the complete measurement, so its size is guarded here on a canned measurement
of every workload the default run prints (the round-5 line of record plus the four `*_cutils` workloads added in round 6)."""
import copy
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned_measurement():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))     # a COMPLETE measurement (the old, long format)
    for name, _ in bench.CUTILS_WORKLOADS:
        w = copy.deepcopy(d["workloads"]["cfg5_d3_dmrebuild"])            # (the largest of the sub-objects)
        w["kernel_ms"] = {"step": 0.0361234567, "obs_cutils": 0.81234567, "policy_pack": 0.0123456789}
        d["workloads"]["%s_cutils" % name] = w
    return d


def expected_workload_keys():
    keys = ["%s_d%d%s" % (n, dep, "_dmrebuild" if rb else "") for n, dep, rb, _ in bench.EXTRA_WORKLOADS]
    keys += ["%s_d%d%s%s" % (n, dep, "_distinct%d" % k if k else "", "_spfollow" if kind == 2 else "") for n, dep, rb, _, k, kind in bench.REALISM_WORKLOADS]
    keys += ["%s_d%d%s_keeprows" % (n, dep, "_dmrebuild" if rb else "") for n, dep, rb, _ in bench.KEEP_ROWS_WORKLOADS]
    keys += ["%s_cutils" % n for n, _ in bench.CUTILS_WORKLOADS]
    return keys


def test_compact_line_fits_and_round_trips():
    full = canned_measurement()
    assert sorted(full["workloads"]) == sorted(expected_workload_keys())       # the canned set IS the default run's set
    full["detail_file"] = "gpurun_out/bench_detail.json"
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) < bench.LINE_LIMIT == 8192, len(line)
    d = json.loads(line)
    # the measurement contract's keys, in full
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "workloads", "detail_file"):
        assert k in d, k
    assert d["config"]["workload"].startswith("cfg2") and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and r["traffic"] > r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    # the numbers survive the rounding (6 significant digits on the headline, 4 on the sub-workloads)
    assert abs(d["value"] / full["value"] - 1) < 1e-5 and abs(d["ms_per_step"] / full["ms_per_step"] - 1) < 1e-5
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 256 * 20) < 0.1
    for k, w in d["workloads"].items():
        f = full["workloads"][k]
        assert set(w) == {"value", "ms_per_step", "steps", "envs", "kernel_ms", "launch_class", "roofline"}, k
        assert abs(w["value"] / f["value"] - 1) < 1e-3 and set(w["roofline"]) == {"frac", "traffic_ratio"}
    # the long provenance strings appear once
    assert line.count("stored PMC measurement") == 1 and "roofline.note" in d["notes"]


def test_compact_line_degrades_instead_of_overflowing():
    """many more sub-workloads than the default run has: the line drops prose, then per-kernel times, and still parses"""
    full = canned_measurement()
    for i in range(40):
        full["workloads"]["extra_%d" % i] = copy.deepcopy(full["workloads"]["cfg3_d3"])
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and len(d["workloads"]) == len(full["workloads"])


def test_a_failed_rank_takes_its_siblings_down():
    """bench.self_launch's wait: one rank exits non-zero before the rendezvous -> the others are terminated, not left waiting"""
    sleeper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"])
    failing = subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(3)"])
    t0 = time.time()
    rcs = bench.wait_ranks([sleeper, failing], poll_s=0.05, grace_s=5.0)
    assert time.time() - t0 < 20 and rcs[1] == 3 and rcs[0] not in (None, 0)
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(2)]
    assert bench.wait_ranks(ok, poll_s=0.05) == [0, 0]
