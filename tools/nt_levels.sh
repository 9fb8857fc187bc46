#!/usr/bin/env bash
# time and WRITE_SIZE / FETCH_SIZE of the observation kernel for builds with different OBS_NT_LEVEL (ab_libs/libfl_nt<k>.so; "-" = tree)
set -uo pipefail
mkdir -p gpurun_out/nt
for spec in "cfg2 2" "cfg3 3" "cfg4 2" "cfg5 3 --dm-rebuild"; do
  read wl depth extra <<< "$spec"
  steps=100; [ "$wl" = "cfg2" ] && steps=300; [ "$wl" = "cfg5" ] && steps=60
  for lib in ab_libs/libfl_nt0.so ab_libs/libfl_nt1.so ab_libs/libfl_nt2.so -; do
    name=$(basename $lib .so); arg="--lib $PWD/$lib"; [ "$lib" = "-" ] && { name=tree; arg=""; }
    python bench.py --no-extra-workloads --no-cpu-baseline --workload $wl --tree-depth $depth --steps $steps --warmup 20 $extra $arg 2>/dev/null > gpurun_out/nt/b.json
    v=$(python -c "import json;d=json.load(open('gpurun_out/nt/b.json'));print('%.2f M obs %.4f ms' % (d['value']/1e6, list(d['kernel_ms'].values())[1]))")
    for c in WRITE_SIZE FETCH_SIZE; do
      python tools/pmc_pass.py gpurun_out/nt/p.json $c --no-extra-workloads --workload $wl --tree-depth $depth --steps $steps --warmup 20 $extra $arg > /dev/null 2>&1
      eval "$c=$(python -c "import json;d=json.load(open('gpurun_out/nt/p.json'));print(round([v['$c'] for k,v in d.items() if 'k_obs' in k][0]/1024,1))")"
    done
    echo "$wl $name: $v  WRITE_SIZE $WRITE_SIZE MB  FETCH_SIZE $FETCH_SIZE MB"
  done
done
