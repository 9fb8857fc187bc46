#!/usr/bin/env bash
# A/B builds of the C-ABI library beside the in-tree one:  tools/build_variant.sh NAME [extra hipcc flags...]
#   -> ab_libs/libfl_NAME.so (git-ignored, NOT gpurun-ignored: the library travels to the GPU box; delete ab_libs/ when the experiment is
#      over) with its objects in build_ab/obj_NAME (git- and gpurun-ignored: they stay here)
#   tools/build_variant.sh timing -DFL_OBS_TIMING          (phase clocks: tools/obs_phase_clocks.py)
#   tools/build_variant.sh base                            (a copy of the current sources as the baseline of an A/B run)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
mkdir -p "$ROOT/build_ab" "$ROOT/ab_libs"
OUT="$ROOT/ab_libs/libfl_$name.so" OBJDIR="$ROOT/build_ab/obj_$name" EXTRA_HIPCC_FLAGS="$*" "$ROOT/flatland_marl_amd/csrc/build.sh"
