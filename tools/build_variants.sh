#!/bin/bash
# Dev tool: build libgemini_hip.so with each Fq representation (GM_FQ30 = 0, 1, 2) into tools/_build/var<k>/
# for A/B runs through GM_LIB_PATH.  Usage: tools/build_variants.sh [variants...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in "${@:-0 1 2}"; do
  for k in $v; do
    d=$ROOT/tools/_build/var$k
    rm -rf "$d" && mkdir -p "$d/gemini_amd" "$d/include"
    cp -r "$ROOT/gemini_amd/csrc" "$d/gemini_amd/" && cp "$ROOT"/include/* "$d/include/"
    rm -f "$d"/gemini_amd/csrc/*.o
    make -s -C "$d/gemini_amd/csrc" -j4 GM_FQ30=${GM_FQ30:-2} CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -DGM_FQ30=${GM_FQ30:-2} $EXTRA" >/dev/null
    echo "built $d/gemini_amd/libgemini_hip.so (EXTRA=$EXTRA)"
  done
done
