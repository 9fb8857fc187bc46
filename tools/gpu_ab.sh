#!/usr/bin/env bash
# same-box A/B of builds of the C-ABI library:  [WL="cfg3 3"] tools/gpu_ab.sh TAG lib1.so lib2.so ...   ("-" = the in-tree build)
set -euo pipefail
mkdir -p gpurun_out
tag=$1; shift
set -- "$@"
read wl depth <<< "${WL:-cfg2 2}"
steps=600; [ "$wl" != "cfg2" ] && steps=100
for rep in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so); arg="--lib $lib"; [ "$lib" = "-" ] && { name=tree; arg=""; }
    python bench.py --no-extra-workloads --no-cpu-baseline --workload $wl --tree-depth $depth --steps $steps $arg > gpurun_out/ab_${tag}_$name.json 2> gpurun_out/ab_${tag}_$name.err
    python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], '%.2f M' % (d['value']/1e6), d['kernel_ms'])" gpurun_out/ab_${tag}_$name.json
  done
done
