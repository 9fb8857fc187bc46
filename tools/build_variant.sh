#!/usr/bin/env bash
# A/B builds of the C-ABI library beside the in-tree one:  tools/build_variant.sh NAME [extra hipcc flags...]
#   -> build_ab/libfl_NAME.so (objects in build_ab/obj_NAME; build_ab/ is git-ignored, removed before a round ends and listed in
#      .gpurunignore then: take that line out again to ship A/B libraries to the GPU box)
#   tools/build_variant.sh timing -DFL_OBS_TIMING          (phase clocks: tools/obs_phase_clocks.py)
#   tools/build_variant.sh base                            (a copy of the current sources as the baseline of an A/B run)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
mkdir -p "$ROOT/build_ab"
OUT="$ROOT/build_ab/libfl_$name.so" OBJDIR="$ROOT/build_ab/obj_$name" EXTRA_HIPCC_FLAGS="$*" "$ROOT/flatland_marl_amd/csrc/build.sh"
