"""Profiles of record (run on the GPU box): for every bench workload, one `rocprofv3 --kernel-trace --stats` run and two
PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) over

    python3 bench.py --no-cpu-baseline --no-extra-workloads --workload W --tree-depth D [--dm-rebuild] --steps N

Writes gpurun_out/prof_<tag>/: <tag>_<W>_d<D>_kernel_stats.csv (copied verbatim from rocprofv3), <tag>_<W>_d<D>_bench.json,
and pmc_traffic.json (per workload and kernel: mean FETCH_SIZE / WRITE_SIZE per launch, HBM bytes per launch =
2 * FETCH_SIZE + WRITE_SIZE KiB -- on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads; the factor is an
upper bound for narrow gathers -- stamped with the sha of the kernel sources).  Copy what is to be judged into profiles/.

  python tools/profile_workloads.py r02 [cfg2:2 cfg3:3 cfg4:2 cfg5:3:rebuild cfg4:2:distinct4 cfg5:3:rebuild:distinct2]
(":distinctK": K distinct generated maps instead of the committed base envs -- key <W>_d<D>_distinctK; ":pack" with depth 0: the flatland_cutils
builder alone writing the policy's tensors -- key <W>_d0; ":keeprows": FL_OBS_KEEP_TREE_ROWS -- key ..._keeprows)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha only; nothing touches the GPU at import)

TABLES = {"k_step<": "k_step<synth>", "k_distance_map": "k_distance_map", "k_hop8": "k_hop8", "k_nexthop": "k_nexthop", "k_segments": "k_segments", "k_policy_pack": "k_policy_pack"}


def names_of(pack):
    """kernel-name prefix -> the stage name bench.py looks the stored traffic up under"""
    n = dict(TABLES)
    if pack:      # the flatland_cutils builder alone writing the policy's tensors: MODE 0 / 6 / 7 / 8 and the large-map split kernels
        n.update({"k_obs<%d" % m: "k_obs<cutils,i64>" for m in (0, 6, 7, 8)})
        n["k_obs_split<0"] = "k_obs<cutils,i64>"
    else:
        n.update({"k_obs<0": "k_obs<cutils>", "k_obs<1": "k_obs<tree>", "k_obs_split<": "k_obs<cutils+tree>"})
        n.update({"k_obs<%d" % m: "k_obs<cutils+tree>" for m in (2, 3, 4, 5)})
    return n


NAMES = names_of(False)
tag = sys.argv[1]
specs = sys.argv[2:] or ["cfg2:2", "cfg3:3", "cfg4:2", "cfg5:3:rebuild"]
out_dir = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
traffic = {}
sha = bench.kernel_source_sha()


def run(args, bench_args, tmp):
    shutil.rmtree(tmp, ignore_errors=True)
    cmd = ["rocprofv3"] + args + ["-d", tmp, "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args
    p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if p.returncode != 0:
        print("FAILED:", " ".join(cmd), p.stderr[-400:], flush=True)
    return p


def pmc_mean(tmp, counter, NAMES):
    acc, n = collections.defaultdict(list), collections.Counter()
    for path in glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            for k, v in NAMES.items():
                if k in row["Kernel_Name"]:
                    acc[v].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


for spec in specs:
    parts = spec.split(":")
    w, depth, rebuild = parts[0], int(parts[1]), "rebuild" in parts[2:]
    distinct = next((int(x[len("distinct"):]) for x in parts[2:] if x.startswith("distinct")), 0)
    pack, keeprows = "pack" in parts[2:], "keeprows" in parts[2:]      # (pack: depth 0, fl_obs_cutils_policy; keeprows: FL_OBS_KEEP_TREE_ROWS)
    NAMES = names_of(pack)
    key = "%s_d%d" % (w, depth) + ("_distinct%d" % distinct if distinct else "") + ("_keeprows" if keeprows else "")
    steps = {"cfg2": 300, "cfg3": 100, "cfg4": 100, "cfg5": 60}[w]
    bargs = ["--no-cpu-baseline", "--no-extra-workloads", "--workload", w, "--tree-depth", str(depth), "--steps", str(steps), "--warmup", "20"]
    if rebuild:
        bargs.append("--dm-rebuild")
    if distinct:
        bargs += ["--distinct-maps", str(distinct)]
    if pack:
        bargs += ["--pack", "1"]
    if keeprows:
        bargs.append("--keep-rows")
    tmp = "/tmp/prof_%s_%d" % (key, os.getpid())
    p = run(["--kernel-trace", "--stats", "--output-format", "csv"], bargs, tmp)
    stats = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, key)))
    try:
        json.dump(json.loads(p.stdout.strip().splitlines()[-1]), open(os.path.join(out_dir, "%s_%s_bench.json" % (tag, key)), "w"), indent=1)
    except Exception:
        pass
    run(["--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv"], bargs, tmp)
    f, nf = pmc_mean(tmp, "FETCH_SIZE", NAMES)
    run(["--pmc", "WRITE_SIZE", "--kernel-trace", "--output-format", "csv"], bargs, tmp)
    wv, nw = pmc_mean(tmp, "WRITE_SIZE", NAMES)
    shutil.rmtree(tmp, ignore_errors=True)
    import flatland_marl_amd.workload as wl
    traffic[key] = {k: dict(envs=wl.WORKLOADS[w]["B"], tag=tag, kernel_source_sha=sha, launches=min(nf[k], nw.get(k, 0)),
                            fetch_size_kib=f[k], write_size_kib=wv.get(k, 0.0), hbm_bytes_per_launch=(2 * f[k] + wv.get(k, 0.0)) * 1024)
                    for k in f}
    print(key, {k: (round(v["fetch_size_kib"]), round(v["write_size_kib"])) for k, v in traffic[key].items()}, flush=True)
    kept = os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, key))
    if os.path.exists(kept):
        for row in list(csv.DictReader(open(kept)))[:6]:
            print("   ", row.get("Name", "")[:40], row.get("Calls"), row.get("AverageNs"), flush=True)
# (several GPU-box calls fill one directory -- a call is limited to 20 minutes --, each with its own file: tools/install_artefacts.py merges them)
json.dump(traffic, open(os.path.join(out_dir, "pmc_traffic%s.json" % ("_" + os.environ["PART"] if os.environ.get("PART") else "")), "w"), indent=1, sort_keys=True)
