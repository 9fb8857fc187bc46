#!/usr/bin/env bash
# same-box sweep of environment settings on one workload:  tools/gpu_env_sweep.sh TAG "WORKLOAD DEPTH STEPS" "VAR=1" "VAR=2 OTHER=x" ...   ("-" = none)
set -euo pipefail
mkdir -p gpurun_out
tag=$1; read wl depth steps <<< "$2"; shift 2
extra=""; [ "$wl" = "cfg5" ] && extra="--dm-rebuild"
n=0
for setting in "$@"; do
  n=$((n+1)); [ "$setting" = "-" ] && setting=""
  env $setting FL_OBS_VERBOSE=1 python bench.py --no-extra-workloads --no-cpu-baseline --workload $wl --tree-depth $depth --steps $steps --warmup 20 $extra > gpurun_out/sw_${tag}_$n.json 2> gpurun_out/sw_${tag}_$n.err || { echo "$setting FAILED"; tail -3 gpurun_out/sw_${tag}_$n.err; continue; }
  python -c "
import json,sys;d=json.load(open(sys.argv[1]));print('%-12s %-60s %.2f M  obs %.4f ms' % (sys.argv[3], sys.argv[2] or '(default)', d['value']/1e6, d['kernel_ms'].get('obs_cutils_tree_fused', 0)))" gpurun_out/sw_${tag}_$n.json "$setting" "$wl-d$depth"
  grep -m1 "^\[fl_obs\]" gpurun_out/sw_${tag}_$n.err | cut -c1-220 || true
done
