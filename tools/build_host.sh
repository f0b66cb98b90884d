#!/bin/bash
# Build the UNMODIFIED reference host (SoilMachine.cpp) against the soilmx compat headers, headless.
#
# GCC resolves quoted includes relative to the directory the including file was NAMED by, so a symlink
# build/host/SoilMachine.cpp -> $REF_ROOT/SoilMachine.cpp makes its `#include "source/layermap.h"` etc. pick up
# build/host/source -> soilmachine_amd/compat/source (our drop-in headers) instead of the reference's own.
# <TinyEngine/...> and <glm/glm.hpp> come from tools/hoststub (neither is vendored by the reference); nothing under
# oracle/ is on this build's include path.
# Needs $REF_ROOT (this container only); the binary links libsoilmx.so and therefore needs a GPU to RUN.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF_ROOT="${REF_ROOT:-/root/reference}"
[ -f "$REF_ROOT/SoilMachine.cpp" ] || { echo "no reference tree at $REF_ROOT"; exit 0; }
OUT="$ROOT/build/host"
mkdir -p "$OUT"
ln -sfn "$REF_ROOT/SoilMachine.cpp" "$OUT/SoilMachine.cpp"
ln -sfn "$ROOT/soilmachine_amd/compat/source" "$OUT/source"
g++ -std=c++20 -O2 -ffp-contract=off -Wfatal-errors -w \
    -I"$ROOT/tools/hoststub" -I"$ROOT/include" \
    -o "$OUT/soilmachine_headless" "$OUT/SoilMachine.cpp" \
    -L"$ROOT/soilmachine_amd" -lsoilmx -Wl,-rpath,'$ORIGIN/../../soilmachine_amd'
echo "built $OUT/soilmachine_headless"
