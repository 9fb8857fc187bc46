#!/usr/bin/env bash
# builds and runs the FETCH_SIZE / WRITE_SIZE calibration (on the GPU box):  tools/pmc_calibrate/run.sh  -> gpurun_out/pmc_calibration.json
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$ROOT/gpurun_out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$HERE/calib.hip" -o /tmp/fl_calib
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/calib_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/calib_$c -- /tmp/fl_calib > /tmp/calib_$c.log 2>&1 || { tail -5 /tmp/calib_$c.log; exit 1; }
done
python3 - "$ROOT/gpurun_out/pmc_calibration.json" <<'PY'
import collections, csv, glob, json, sys
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for path in glob.glob("/tmp/calib_%s/**/*_counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == c:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    out[c] = {k: sum(v) / len(v) for k, v in acc.items()}
GiB, n_acc = 1 << 30, 1 << 24
exp = {"k_stream_read": ("FETCH_SIZE", GiB), "k_gather_read": ("FETCH_SIZE", n_acc * 64), "k_gather16_read": ("FETCH_SIZE", n_acc * 64),
       "k_stream_write": ("WRITE_SIZE", GiB), "k_scatter_write": ("WRITE_SIZE", n_acc * 64)}
res = {}
for k, (c, b) in exp.items():
    kib = out[c].get(k, float("nan"))
    res[k] = dict(counter=c, reported_kib=kib, reported_bytes=kib * 1024, bytes_touched_at_line_granularity=b, reported_over_touched=kib * 1024 / b)
    print("%-16s %-10s reported %.1f MiB, lines touched %.1f MiB -> ratio %.3f" % (k, c, kib / 1024, b / 2**20, kib * 1024 / b))
json.dump(dict(note="rocprofv3 PMC on gfx950; *_SIZE counters are KiB; ratio = reported bytes / (lines touched x 64 B)", kernels=res, raw=out), open(sys.argv[1], "w"), indent=1)
PY
