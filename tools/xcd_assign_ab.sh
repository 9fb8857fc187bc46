set -o pipefail
mkdir -p gpurun_out/xcd
for K in 2 8; do
 for A in mod block; do
  FL_MAP_ASSIGN=$A timeout -k 10 300 python bench.py --workload cfg5 --tree-depth 3 --dm-rebuild --distinct-maps $K --steps 40 --warmup 8 --no-cpu-baseline --no-extra-workloads > gpurun_out/xcd/cfg5_K${K}_$A.json 2> gpurun_out/xcd/cfg5_K${K}_$A.err || exit 1
  python - <<PY
import json
r=json.loads(open("gpurun_out/xcd/cfg5_K${K}_$A.json").read().strip().splitlines()[-1])
print("cfg5 K=$K $A", r["value"], r["ms_per_step"], r["config"].get("distinct_maps"))
PY
 done
done
for A in mod block; do
  FL_MAP_ASSIGN=$A timeout -k 10 300 python bench.py --workload cfg4 --tree-depth 2 --distinct-maps 4 --steps 100 --warmup 8 --no-cpu-baseline --no-extra-workloads > gpurun_out/xcd/cfg4_K4_$A.json 2> gpurun_out/xcd/cfg4_K4_$A.err || exit 1
  python - <<PY
import json
r=json.loads(open("gpurun_out/xcd/cfg4_K4_$A.json").read().strip().splitlines()[-1])
print("cfg4 K=4 $A", r["value"], r["ms_per_step"])
PY
done
