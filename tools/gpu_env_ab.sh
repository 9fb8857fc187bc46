#!/usr/bin/env bash
# same-box A/B of environment settings on the cfg2 bench line:  tools/gpu_env_ab.sh TAG "VAR=1" "VAR=2 OTHER=x" ...   ("-" = none)
set -euo pipefail
mkdir -p gpurun_out
tag=$1; shift
for rep in 1 2; do
  n=0
  for setting in "$@"; do
    n=$((n+1)); [ "$setting" = "-" ] && setting=""
    env $setting python bench.py --no-extra-workloads --no-cpu-baseline > gpurun_out/env_${tag}_$n.json 2> gpurun_out/env_${tag}_$n.err
    python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2] or '(default)', '%.2f M' % (d['value']/1e6), d['kernel_ms'])" gpurun_out/env_${tag}_$n.json "$setting"
  done
done
