#!/usr/bin/env bash
# same-box A/B over several workloads:  tools/gpu_ab_wl.sh TAG "cfg4 2 0,cfg4 2 4,cfg3 3 0" lib1.so lib2.so ...   ("-" = the in-tree build)
# (workload, tree depth, distinct maps); two repetitions, alternating the libraries
set -euo pipefail
mkdir -p gpurun_out
tag=$1; wls=$2; shift 2
IFS=',' read -ra WLS <<< "$wls"
for rep in 1 2; do
  for spec in "${WLS[@]}"; do
    read wl depth distinct <<< "$spec"
    steps=600; [ "$wl" != "cfg2" ] && steps=100
    extra=""; [ "$distinct" != "0" ] && extra="--distinct-maps $distinct"; [ "$wl" = "cfg5" ] && extra="$extra --dm-rebuild"
    for lib in "$@"; do
      name=$(basename $lib .so); arg="--lib $lib"; [ "$lib" = "-" ] && { name=tree; arg=""; }
      python bench.py --no-extra-workloads --no-cpu-baseline --workload $wl --tree-depth $depth --steps $steps --warmup 20 $extra $arg > gpurun_out/ab_${tag}_$name.json 2> gpurun_out/ab_${tag}_$name.err || { tail -3 gpurun_out/ab_${tag}_$name.err; continue; }
      python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2], sys.argv[3], '%.2f M' % (d['value']/1e6), d['kernel_ms'], d.get('launch_class'))" gpurun_out/ab_${tag}_$name.json "$spec" $name
    done
  done
done
