"""Diagnostic: one rocprofv3 --pmc pass over `python3 bench.py ...` and the per-kernel means of the counters.

  python tools/pmc_pass.py OUT.json "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." [bench.py args...]        (on the GPU box)

Counters are collected in their own run (with --kernel-trace only), as MI355X_MICROARCH.md prescribes."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_json, counters, bench_args = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
tmp = "/tmp/pmc_pass_%d" % os.getpid()
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "--",
                                            "python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + bench_args
subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"][:24]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k in res:
    res[k]["launches"] = max(len(v) for v in acc[k].values())
json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)
shutil.rmtree(tmp, ignore_errors=True)
print(json.dumps({k: v for k, v in res.items() if "k_obs" in k or "k_step" in k}, indent=1, sort_keys=True))
