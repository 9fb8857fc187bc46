#!/usr/bin/env bash
# Builds the C-ABI shared library for gfx950 in-tree (travels to the GPU box with the snapshot).
# One object per translation unit, compiled in parallel; an object is rebuilt when its source, a header, this script or the
# flags changed (the flags' hash is part of the stamp, so a flags-only change rebuilds on a box that already has the .so).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${OUT:-$HERE/libflatland_hip.so}"
OBJDIR="${OBJDIR:-$HERE/build}"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
UNITS=(fl_obs_s4b fl_obs_f21 fl_obs_f16 fl_obs_f20 fl_obs_f14 fl_obs_s9b fl_obs_f19 fl_obs_f15 fl_obs_f13 fl_obs_f12 fl_obs_f18 fl_obs_f17 fl_obs_f11 fl_obs_f5 fl_obs_f10 fl_obs_f9 fl_obs_s9 fl_obs_f8 fl_obs_f7 fl_obs_m7 fl_obs_m8 fl_obs_m6 fl_obs_f6 fl_obs_s4 fl_obs_s3 fl_obs_s2 fl_obs_m2 fl_obs_m4 fl_obs_m5 fl_obs_f4 fl_obs_f3 fl_obs_f2 fl_obs_m3 fl_obs_m0 fl_obs_m1 fl_obs_f1 fl_host fl_step fl_dmap fl_obs)
# -disable-machine-licm: the observation kernel sits at its 128-VGPR / 102-SGPR ceiling (1024 threads a workgroup); hoisting
# loop invariants out of the loops over the rounds of trees only adds spills (same-box A/B: cfg3 / cfg4 / cfg5 2.4 - 3.2 % faster
# without it, cfg2 unchanged)
# -amdgpu-atomic-optimizer-strategy=None: the atomics on one address are issued by ONE lane for its wavefront already (work-list
# and queue counters); the optimizer's own wave reduction around them only adds instructions and waits (k_step 12.6 -> 12.0 us)
# -disable-lsr: loop strength reduction turns the loops' index arithmetic into extra induction registers; at the register ceiling
# that is spills and moves (same-box A/B: cfg2 k_obs 52.3 -> 50.9 us, cfg3 0.777 -> 0.765 ms, cfg4 / cfg5 unchanged)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -disable-machine-licm ${FL_LSR_FLAGS--mllvm -disable-lsr}
       -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-result ${EXTRA_HIPCC_FLAGS:-})
mkdir -p "$OBJDIR"
# stamp = hash of everything every unit depends on besides its own source: headers, this script, the flags
stamp=$( (cat "$HERE"/*.h "$HERE/../../include/flatland_hip.h" "$0"; echo "${FLAGS[*]}") | sha256sum | cut -c1-16)
pids=()
rebuilt=0
for u in "${UNITS[@]}"; do
  src="$HERE/$u.hip"; obj="$OBJDIR/$u.o"; tag="$OBJDIR/$u.stamp"
  want="$stamp $(sha256sum < "$src" | cut -c1-16)"
  if [ -f "$obj" ] && [ -f "$tag" ] && [ "$(cat "$tag")" = "$want" ] && [ -z "${FORCE:-}" ]; then continue; fi
  rebuilt=1
  ( "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$obj" && echo "$want" > "$tag" ) &
  pids+=($!)
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
if [ "$rebuilt" = 0 ] && [ -f "$OUT" ] && [ -f "$OBJDIR/link.stamp" ] && [ "$(cat "$OBJDIR/link.stamp")" = "$stamp" ]; then
  echo "up to date: $OUT"; exit 0
fi
objs=(); for u in "${UNITS[@]}"; do objs+=("$OBJDIR/$u.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "$stamp" > "$OBJDIR/link.stamp"
echo "built $OUT"
