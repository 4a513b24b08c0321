#!/bin/bash
# Measurement build of the engine with the in-kernel phase timers (k_fine: -DVELLO_FINE_PROF, k_coarse: -DVELLO_COARSE_PROF)
# -> ab_tmp/libvello_hip_PROF.so (git-ignored, travels with gpurun); the in-tree product library is not touched.
# OUT_NAME=<X> PROF_FLAGS="-D..." builds any other variant of the in-tree sources into ab_tmp/libvello_hip_<X>.so.
# Then on the GPU box:  python scripts/fine_prof.py [d2|r1mix]   /   python scripts/coarse_prof.py
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT_NAME="${OUT_NAME:-PROF}"
W="${TMPDIR:-/tmp}/vello_prof_build_$OUT_NAME"
rm -rf "$W" && mkdir -p "$W/vello_amd" "$ROOT/ab_tmp"
cp -r "$ROOT/vello_amd/csrc" "$W/vello_amd/" && cp -r "$ROOT/include" "$W/"
rm -rf "$W/vello_amd/csrc/build"
make -s -j8 -C "$W/vello_amd/csrc" EXTRA="${PROF_FLAGS:--DVELLO_FINE_PROF -DVELLO_COARSE_PROF}"
cp "$W/vello_amd/lib/libvello_hip.so" "$ROOT/ab_tmp/libvello_hip_$OUT_NAME.so"
ls -la "$ROOT/ab_tmp/libvello_hip_$OUT_NAME.so"
