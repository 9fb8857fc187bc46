"""Copies a round's artefacts from gpurun_out/prof_<tag>/ into profiles/ (the bench line of record is installed separately, after
the traffic entries are in place) and prints the summary the docs quote.   python tools/install_artefacts.py r04"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", "prof_" + tag), os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(src, tag + "_*")):
    shutil.copy(f, dst)      # (incl. the default bench line + its detail file, when PART=rest has run)
t = json.load(open(os.path.join(dst, "pmc_traffic.json")))
new = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_traffic*.json"))):     # one file per GPU-box call (PART=traces1 / traces2 / traces3)
    new.update(json.load(open(f)))
t.update(new)
json.dump(t, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def obs_entry(e):
    return next(v for k, v in e.items() if k.startswith("k_obs<"))


print("tree sha", bench.kernel_source_sha(), "artefacts", {obs_entry(new[k])["kernel_source_sha"] for k in new})
for w in sorted(new):
    rows = list(csv.DictReader(open(os.path.join(src, "%s_%s_kernel_stats.csv" % (tag, w)))))
    o = [r for r in rows if "k_obs" in r["Name"]][0]
    s = [r for r in rows if "k_step" in r["Name"]][0]
    b = json.load(open(os.path.join(src, "%s_%s_bench.json" % (tag, w))))
    e = obs_entry(new[w])
    alg, us = b["roofline"]["algorithmic_bytes_per_launch"], float(o["AverageNs"]) / 1e3
    print("%-22s %-16s trace %7.1f us  step %5.1f us  %6.1f M  %5.0f GB/s (%.3f)  F %7.1f W %7.1f MB  2F+W %.2f  F+W %.2f" % (
        w, o["Name"][5:21], us, float(s["AverageNs"]) / 1e3, b["value"] / 1e6, alg / us / 1e3, alg / us / 1e3 / 8000,
        e["fetch_size_kib"] * 1024 / 1e6, e["write_size_kib"] * 1024 / 1e6, (2 * e["fetch_size_kib"] + e["write_size_kib"]) * 1024 / alg,
        (e["fetch_size_kib"] + e["write_size_kib"]) * 1024 / alg))
sqp = os.path.join(src, "%s_sq_counters_cfg2.json" % tag)
if os.path.exists(sqp):
    for k, v in json.load(open(sqp)).items():
        if "k_obs" in k:
            print({c: round(x) for c, x in v.items()})
dp = os.path.join(src, "%s_bench_default.json" % tag)
if os.path.exists(dp):
    d = json.load(open(dp))
    print("default run:", round(d["value"] / 1e6, 2), {k: round(v["value"] / 1e6, 1) for k, v in d.get("workloads", {}).items()})
