#!/usr/bin/env bash
# One GPU-box cycle while tuning the observation kernel: parity tests, the cfg2 bench line (A/B against ab_libs/libfl_base.so
# on the same box when that build exists), the phase clocks of the timing build.
#   tools/gpu_cycle.sh TAG [workload depth]       (run through gpurun; writes gpurun_out/{t,b,clk}_TAG.*)
set -euo pipefail
mkdir -p gpurun_out
tag=$1; wl=${2:-cfg2}; depth=${3:-2}
if [ -z "${SKIP_TESTS:-}" ]; then python -m pytest tests -m gpu -x -q > gpurun_out/t_$tag.log 2>&1; tail -3 gpurun_out/t_$tag.log; fi
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], '%.2f M' % (d['value']/1e6), d['kernel_ms'])" $1; }
for rep in 1 2; do
  if [ -f ab_libs/libfl_base.so ]; then
    python bench.py --no-extra-workloads --lib ab_libs/libfl_base.so > gpurun_out/b_${tag}_base.json 2> gpurun_out/b_${tag}_base.err; show gpurun_out/b_${tag}_base.json
  fi
  python bench.py --no-extra-workloads > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err; show gpurun_out/b_$tag.json
done
if [ -f ab_libs/libfl_timing.so ]; then python tools/obs_phase_clocks.py ab_libs/libfl_timing.so $wl $depth > gpurun_out/clk_$tag.txt 2>&1; cat gpurun_out/clk_$tag.txt; fi
