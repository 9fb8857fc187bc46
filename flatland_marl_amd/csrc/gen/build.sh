#!/usr/bin/env bash
# Builds the host-side generator library (plain C++17, no GPU) in-tree.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${OUT:-$HERE/libflatland_gen.so}"
SRC="$HERE/fl_gen.cpp"
HDR="$HERE/../../../include/flatland_gen.h"
if [ -f "$OUT" ] && [ "$OUT" -nt "$SRC" ] && [ "$OUT" -nt "$HDR" ] && [ -z "${FORCE:-}" ]; then echo "up to date: $OUT"; exit 0; fi
g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function "$SRC" -o "$OUT"
echo "built $OUT"
