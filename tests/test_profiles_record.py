"""CPU: the profiles of record under profiles/ are complete and consistent with each other -- every workload has its rocprofv3
kernel summary, its bench line and its PMC traffic entry from ONE build (the same kernel-source stamp), and the default bench
line carries what the measurement contract asks for -- and that build IS the tree's: the stamp of the stored PMC traffic equals
the hash of the kernel sources here (a kernel change without new profiles of record fails this test; bench.py reports
`roofline.traffic` from the stored measurement only under the same condition)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
# (workload key, the stage the bench line prices: both builders in one launch / the flatland_cutils builder alone writing the policy's tensors)
WORKLOADS = (("cfg2_d2", "k_obs<cutils+tree>"), ("cfg3_d3", "k_obs<cutils+tree>"), ("cfg4_d2", "k_obs<cutils+tree>"), ("cfg5_d3", "k_obs<cutils+tree>"),
             ("cfg2_d2_distinct10", "k_obs<cutils+tree>"), ("cfg3_d3_distinct10", "k_obs<cutils+tree>"), ("cfg4_d2_distinct4", "k_obs<cutils+tree>"),
             ("cfg5_d3_distinct2", "k_obs<cutils+tree>"),                         # distinct generated maps
             ("cfg2_d0", "k_obs<cutils,i64>"), ("cfg3_d0", "k_obs<cutils,i64>"), ("cfg4_d0", "k_obs<cutils,i64>"), ("cfg5_d0", "k_obs<cutils,i64>"),   # the consumer's path
             ("cfg3_d3_keeprows", "k_obs<cutils+tree>"), ("cfg5_d3_keeprows", "k_obs<cutils+tree>"))     # FL_OBS_KEEP_TREE_ROWS
TAG = "r06"


def test_every_workload_has_trace_bench_line_and_traffic_from_one_build():
    traffic = json.load(open(os.path.join(P, "pmc_traffic.json")))
    stamps = set()
    for w, stage in WORKLOADS:
        rows = list(csv.DictReader(open(os.path.join(P, f"{TAG}_{w}_kernel_stats.csv"))))
        obs = [r for r in rows if "k_obs" in r["Name"]]
        assert len(obs) == 1 and int(obs[0]["Calls"]) >= 100 and float(obs[0]["AverageNs"]) > 0
        line = json.load(open(os.path.join(P, f"{TAG}_{w}_bench.json")))
        assert line["metric"] == "agent_steps_per_sec" or "agent" in line["metric"]
        assert line["roofline"]["algorithmic_bytes_per_launch"] > 0
        # the event-timed launch of the bench line and the trace average of the same run agree (events add a few microseconds)
        ev_ms, tr_ms = line["roofline"]["kernel_ms"], float(obs[0]["AverageNs"]) / 1e6
        assert 0.9 * tr_ms < ev_ms < 1.15 * tr_ms + 0.005, (w, ev_ms, tr_ms)
        t = traffic[w][stage]
        assert line["roofline"]["kernel"] == stage, (w, line["roofline"]["kernel"])
        assert t["tag"] == TAG and t["fetch_size_kib"] > 0 and t["write_size_kib"] > 0
        # the kernel writes at least its outputs (the keep-rows mode's algorithmic bytes count the REAL rows it writes; a row that turns constant is written too)
        assert t["write_size_kib"] * 1024 > 0.9 * line["roofline"]["algorithmic_bytes_per_launch"]
        stamps.add(t["kernel_source_sha"])
    assert len(stamps) == 1, stamps
    # ... and that build is this tree's (the stored traffic is stale otherwise: re-run tools/round_artefacts.sh)
    sys.path.insert(0, ROOT)
    import bench
    assert stamps == {bench.kernel_source_sha()}, "profiles of record are from other kernel sources than the tree's"


def test_default_bench_line_of_record_keeps_the_contract():
    raw = open(os.path.join(P, f"{TAG}_bench_default.json")).read()
    assert len(raw.encode()) < 8192 and raw.count("\n") <= 1          # ONE line the driver can capture whole (round 5's was 24 958 bytes)
    d = json.loads(raw)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["config"]["workload"].startswith("cfg2") and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is not None and r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["cpu_model"] and c["sample"]
    assert c["cores"] == c["physical_cores"] <= c["threads"]      # one worker per PHYSICAL core is the stated figure
    # whole-job throughput and time per step belong together: B * A agent-steps per step
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 256 * 20) < 1.0
    # every sub-workload of the default run, compact; the consumer's path among them; the complete objects in the detail file of record
    w = d["workloads"]
    for k in ("cfg3_d3", "cfg4_d2", "cfg5_d3_dmrebuild", "cfg2_cutils", "cfg3_cutils", "cfg4_cutils", "cfg5_cutils", "cfg3_d3_keeprows"):
        assert w[k]["value"] > 0 and w[k]["roofline"]["frac"] > 0 and w[k]["launch_class"][0] != 0, k
        assert w[k]["roofline"]["traffic_ratio"] is not None and w[k]["roofline"]["traffic_ratio"] >= 0.9, k     # a stored PMC measurement for every one
    full = json.load(open(os.path.join(P, f"{TAG}_bench_detail.json")))
    assert abs(full["value"] / d["value"] - 1) < 1e-5 and set(full["workloads"]) == set(w)


def test_sq_counters_of_record_are_this_tree_s_and_consistent():
    """the SQ counters the bench line's `roofline.valu_issue` quotes: taken on the tree's kernel sources, and the two independent
    counters of the vector instructions agree on 4 cycles a wave64 instruction (ACTIVE_INST_VALU is in quad-cycles)"""
    sys.path.insert(0, ROOT)
    import bench
    sq = json.load(open(os.path.join(P, f"{TAG}_sq_counters_cfg2.json")))
    assert sq["kernel_source_sha"] == bench.kernel_source_sha() and sq["tag"] == TAG
    c = sq[[k for k in sq if k.startswith("void k_obs")][0]]
    assert abs(c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"] - 4.0) < 0.1
    frac = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    assert 0.3 < frac < 1.0, frac
