#!/usr/bin/env bash
# Builds the reference's own native module (flatland_cutils, pybind11/C++17) from the
# sources WHERE THEY LIE under /root/reference into oracle/_ref/ (git-ignored).
# Recipe = flatland_cutils/setup.py:18-26 restated as one g++ line; no reference source is
# copied into this repo.  The module needs the (Python) flatland-rl reference to run, so it
# is only usable in the build container, to capture golden vectors (capture_golden.py).
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
mkdir -p "$OUT"
SRC="$REF/flatland_cutils/src"
if [ ! -d "$SRC" ]; then echo "reference not present; skipping oracle/_ref build"; exit 0; fi
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
TARGET="$OUT/flatland_cutils$EXT"
if [ -f "$TARGET" ] && [ "$TARGET" -nt "$SRC/treeobs.cpp" ]; then echo "up to date: $TARGET"; exit 0; fi
g++ -O2 -std=c++17 -shared -fPIC -w $(python3 -m pybind11 --includes) \
    "$SRC/deadlock_checker.cpp" "$SRC/feature_parser.cpp" "$SRC/loader.cpp" \
    "$SRC/main.cpp" "$SRC/predictions.cpp" "$SRC/treeobs.cpp" -o "$TARGET"
echo "built $TARGET"
