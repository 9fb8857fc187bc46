#!/usr/bin/env bash
# same-box A/B of builds of the C-ABI library on the cfg2 bench line:  tools/gpu_ab.sh TAG lib1.so lib2.so ...   ("-" = the in-tree build)
tag=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so); arg="--lib $lib"; [ "$lib" = "-" ] && { name=tree; arg=""; }
    python bench.py --no-extra-workloads $arg > gpurun_out/ab_${tag}_$name.json 2> gpurun_out/ab_${tag}_$name.err
    python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1], '%.2f M' % (d['value']/1e6), d['kernel_ms'])" gpurun_out/ab_${tag}_$name.json
  done
done
