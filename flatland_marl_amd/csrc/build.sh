#!/usr/bin/env bash
# Builds the C-ABI shared library for gfx950 in-tree (travels to the GPU box with the snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${OUT:-$HERE/libflatland_hip.so}"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS=("$HERE/fl_host.hip" "$HERE/fl_step.hip" "$HERE/fl_dmap.hip" "$HERE/fl_obs.hip")
newest=0
for f in "${SRCS[@]}" "$HERE"/*.h "$HERE/../../include/flatland_hip.h"; do
  m=$(stat -c %Y "$f"); [ "$m" -gt "$newest" ] && newest=$m
done
if [ -f "$OUT" ] && [ "$(stat -c %Y "$OUT")" -ge "$newest" ] && [ -z "${FORCE:-}" ]; then echo "up to date: $OUT"; exit 0; fi
# -disable-machine-licm: the observation kernel sits at its 128-VGPR / 102-SGPR ceiling (1024 threads a workgroup); hoisting
# loop invariants out of the loops over the rounds of trees only adds spills (same-box A/B: cfg3 / cfg4 / cfg5 2.4 - 3.2 % faster
# without it, cfg2 unchanged)
# -amdgpu-atomic-optimizer-strategy=None: the atomics on one address are issued by ONE lane for its wavefront already (work-list
# and queue counters); the optimizer's own wave reduction around them only adds instructions and waits (k_step 12.6 -> 12.0 us)
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -disable-machine-licm \
  -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -Wno-unused-result ${EXTRA_HIPCC_FLAGS:-} "${SRCS[@]}" -o "$OUT"
echo "built $OUT"
