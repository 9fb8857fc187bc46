#!/usr/bin/env bash
# same-box A/B of library builds over the four bench workloads:  tools/gpu_ab_all.sh TAG lib1.so lib2.so ...   ("-" = the in-tree build)
set -euo pipefail
mkdir -p gpurun_out
tag=$1; shift
for spec in "cfg2 2 400" "cfg3 3 100" "cfg4 2 100" "cfg5 3 60"; do
  set -- $spec "${@:1}"; wl=$1; depth=$2; steps=$3; shift 3
  for lib in "$@"; do
    name=$(basename $lib .so); arg="--lib $lib"; [ "$lib" = "-" ] && { name=tree; arg=""; }
    extra=""; [ "$wl" = "cfg5" ] && extra="--dm-rebuild"
    python bench.py --no-extra-workloads --no-cpu-baseline --workload $wl --tree-depth $depth --steps $steps --warmup 20 $extra $arg > gpurun_out/ab_${tag}_${wl}_$name.json 2> gpurun_out/ab_${tag}_${wl}_$name.err || { tail -5 gpurun_out/ab_${tag}_${wl}_$name.err; continue; }
    python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], '%.2f M' % (d['value']/1e6), d['kernel_ms'])" gpurun_out/ab_${tag}_${wl}_$name.json
  done
done
