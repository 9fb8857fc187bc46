#!/usr/bin/env bash
# same-box A/B of environment settings on all four bench workloads (cfg2 d2, cfg3 d3, cfg4 d2, cfg5 d3 + masked rebuild):
#   tools/gpu_env_ab_all.sh TAG "VAR=1" "-" ...      ("-" = none); WLS="cfg3 cfg4" restricts the workloads
set -euo pipefail
mkdir -p gpurun_out
tag=$1; shift
for wl in ${WLS:-cfg2 cfg3 cfg4 cfg5}; do
  case $wl in cfg2) depth=2; steps=600; extra="";; cfg3) depth=3; steps=100; extra="";; cfg4) depth=2; steps=100; extra="";; cfg5) depth=3; steps=60; extra="--dm-rebuild";; *) echo "unknown workload $wl (cfg2 cfg3 cfg4 cfg5)"; continue;; esac
  for rep in 1 2; do
    n=0
    for setting in "$@"; do
      n=$((n+1)); [ "$setting" = "-" ] && setting=""
      env $setting python bench.py --no-extra-workloads --no-cpu-baseline --workload $wl --tree-depth $depth --steps $steps --warmup 20 $extra > gpurun_out/ea_${tag}_${wl}_$n.json 2> gpurun_out/ea_${tag}_${wl}_$n.err || { echo "$wl $setting FAILED"; tail -3 gpurun_out/ea_${tag}_${wl}_$n.err; continue; }
      python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[3], sys.argv[2] or '(default)', '%.2f M' % (d['value']/1e6), d['kernel_ms'])" gpurun_out/ea_${tag}_${wl}_$n.json "$setting" $wl
    done
  done
done
