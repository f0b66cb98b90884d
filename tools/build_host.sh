#!/bin/bash
# Build the UNMODIFIED reference host (SoilMachine.cpp) against the soilmx drop-in headers, headless.
#
# GCC resolves quoted includes relative to the directory the including file was NAMED by, so a symlink
# build/host/SoilMachine.cpp -> $REF_ROOT/SoilMachine.cpp makes its `#include "source/..."` lines pick up
# build/host/source/, which is composed of
#   layermap.h, surface.h, particle/   -> soilmachine_amd/compat/source   (the drop-in: the hot path on the device)
#   io.h, include/, shader/            -> the host tree's OWN files        (.soil parser, exporters, vertex pool, scene,
#                                                                           LBM wind: untouched)
# <TinyEngine/...> and <glm/glm.hpp> (neither is vendored by the reference) and the OpenGL names the host's vertex pool
# uses come from tools/hoststub: window, shaders and draw calls are no-ops, mapped GL buffers are plain host memory.
# Nothing under oracle/ is on this build's include path.
# Needs $REF_ROOT (this container only); the binary links libsoilmx.so and therefore needs a GPU to RUN.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF_ROOT="${REF_ROOT:-/root/reference}"
[ -f "$REF_ROOT/SoilMachine.cpp" ] || { echo "no reference tree at $REF_ROOT"; exit 0; }
OUT="$ROOT/build/host"
rm -rf "$OUT/source"; mkdir -p "$OUT/source"
ln -sfn "$REF_ROOT/SoilMachine.cpp" "$OUT/SoilMachine.cpp"
for f in layermap.h surface.h particle; do ln -sfn "$ROOT/soilmachine_amd/compat/source/$f" "$OUT/source/$f"; done
for f in io.h include shader; do ln -sfn "$REF_ROOT/source/$f" "$OUT/source/$f"; done
g++ -std=c++20 -O2 -ffp-contract=off -Wfatal-errors -w \
    -I"$ROOT/tools/hoststub" -I"$ROOT/include" \
    -o "$OUT/soilmachine_headless" "$OUT/SoilMachine.cpp" \
    -L"$ROOT/soilmachine_amd" -lsoilmx -Wl,-rpath,'$ORIGIN/../../soilmachine_amd'
echo "built $OUT/soilmachine_headless"
# a second host program on the same class API (queries / point modifiers main() never calls): tools/compat_probe.cpp
ln -sfn "$ROOT/tools/compat_probe.cpp" "$OUT/compat_probe.cpp"
g++ -std=c++20 -O2 -ffp-contract=off -Wfatal-errors -w \
    -I"$ROOT/tools/hoststub" -I"$ROOT/include" \
    -o "$OUT/compat_probe" "$OUT/compat_probe.cpp" \
    -L"$ROOT/soilmachine_amd" -lsoilmx -Wl,-rpath,'$ORIGIN/../../soilmachine_amd'
echo "built $OUT/compat_probe"
