#!/usr/bin/env bash
# bench.py on one workload under launcher switches, same box, two repetitions:  tools/gpu_env_ab_bench.sh "<bench args>" "ENV=.." ...   (run through gpurun)
args=$1; shift
for rep in 1 2; do
for sw in "FL_DUMMY=1" "$@"; do
  env $sw python bench.py --no-cpu-baseline --no-extra-workloads $args 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('%-40s %7.2f M  %s  class %s' % ('$sw', d['value']/1e6, {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()}, d['launch_class']))"
done; done
