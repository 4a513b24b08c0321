#!/bin/bash
# The emulator parity suite against a build of the kernel sources with extra compile-time constants -- the way to walk code
# that ordinary scenes never reach (e.g. -DVK_PC_STASH=64 -DVK_PC_TABLE_LOG2=4: k_path_count_agg with a stash of one round and a
# table of 16 entries, so that most crossings take the straight-to-memory arms).  Builds in a scratch copy, swaps the library in
# for the run and puts the tree's own back.      EXTRA="-DVK_PC_STASH=64" bash scripts/emu_variant_check.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SCRATCH="${TMPDIR:-/tmp}/vello_emu_variant"
rm -rf "$SCRATCH" && mkdir -p "$SCRATCH/tests"
cp -r "$ROOT/tests/simt_emu" "$SCRATCH/tests/" && cp -r "$ROOT/vello_amd" "$ROOT/include" "$SCRATCH/"
cd "$SCRATCH/tests/simt_emu" && rm -rf build libvello_emu.so
make -j8 CXXFLAGS="-O2 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mfma -I . -x c++ -Wno-unknown-pragmas ${EXTRA:-}" >/dev/null 2>&1 || true
g++ -shared -o libvello_emu.so build/*.o
cd "$ROOT"
cp tests/simt_emu/libvello_emu.so "$SCRATCH/libvello_emu_orig.so"
trap 'cp "$SCRATCH/libvello_emu_orig.so" "$ROOT/tests/simt_emu/libvello_emu.so"' EXIT
cp "$SCRATCH/tests/simt_emu/libvello_emu.so" tests/simt_emu/libvello_emu.so
python -m pytest tests/test_emu_parity.py -x -q -p no:cacheprovider "$@"
