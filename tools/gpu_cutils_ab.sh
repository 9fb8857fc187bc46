#!/usr/bin/env bash
# cutils-alone launches of one workload under launcher switches (same box):  tools/gpu_cutils_ab.sh cfg4 "FL_OBS_NO_FIX=1" "FL_OBS_FORCE=wl=0" ...
set -uo pipefail
w=$1; shift
mkdir -p gpurun_out/cab
run() { # name, env assignments...
  local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --workload $w --tree-depth 0 --pack 1 --steps ${STEPS:-100} --warmup 20 ${LIBARG:-} --detail gpurun_out/cab/d_${w}_$name.json > gpurun_out/cab/b_${w}_$name.json 2> gpurun_out/cab/b_${w}_$name.err || { tail -3 gpurun_out/cab/b_${w}_$name.err; return; }
  python -c "
import json,sys; d=json.load(open(sys.argv[1])); print('%-40s %7.1f M  %s  class %s' % (sys.argv[2], d['value']/1e6, {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()}, d['launch_class']))" gpurun_out/cab/b_${w}_$name.json "$w $name"
}
run default FL_DUMMY=1
i=0
for sw in "$@"; do i=$((i+1)); run "v$i" $sw; echo "   (v$i = $sw)"; done
run default2 FL_DUMMY=1
