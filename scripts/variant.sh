#!/bin/bash
# Builds an experiment (scripts/experiments/<name>.patch on top of the in-tree kernel sources) without touching the product:
#   scripts/variant.sh <name> <LETTER> [--emu-test [pytest -k expression]]
# -> ab_tmp/libvello_hip_<LETTER>.so (git-ignored, travels with gpurun) for scripts/gpu_ab.sh (VARIANTS="B C ...").
# With --emu-test the patched sources are also compiled for the SIMT emulator, swapped in for tests/simt_emu/libvello_emu.so
# (restored on exit) and tests/test_emu_parity.py is run against them: a variant goes to the GPU only when that is green.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; LETTER=$2; shift 2
W="${TMPDIR:-/tmp}/vello_variant_$LETTER"
rm -rf "$W" && mkdir -p "$W/vello_amd" "$W/tests" "$ROOT/ab_tmp"
cp -r "$ROOT/vello_amd/csrc" "$W/vello_amd/" && cp -r "$ROOT/include" "$W/" && cp -r "$ROOT/tests/simt_emu" "$W/tests/"
rm -rf "$W/vello_amd/csrc/build" "$W/tests/simt_emu/build" "$W/tests/simt_emu/libvello_emu.so"
(cd "$W" && patch -p1 --no-backup-if-mismatch < "$ROOT/scripts/experiments/$NAME.patch")
make -s -j8 -C "$W/vello_amd/csrc"
cp "$W/vello_amd/lib/libvello_hip.so" "$ROOT/ab_tmp/libvello_hip_$LETTER.so"
echo "built ab_tmp/libvello_hip_$LETTER.so from $NAME.patch"
if [ "${1:-}" = "--emu-test" ]; then
  make -s -j8 -C "$W/tests/simt_emu"
  cp "$ROOT/tests/simt_emu/libvello_emu.so" "$W/libvello_emu_orig.so"
  trap 'cp "$W/libvello_emu_orig.so" "$ROOT/tests/simt_emu/libvello_emu.so"' EXIT
  cp "$W/tests/simt_emu/libvello_emu.so" "$ROOT/tests/simt_emu/libvello_emu.so"
  cd "$ROOT"
  if [ -n "${2:-}" ]; then python -m pytest tests/test_emu_parity.py -x -q -p no:cacheprovider -k "$2"; else python -m pytest tests/test_emu_parity.py -x -q -p no:cacheprovider; fi
fi
