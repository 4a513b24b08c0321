#!/bin/bash
# The emulator parity suite against a build of the kernel sources with extra compile-time constants -- the way to walk code
# that ordinary scenes never reach (e.g. -DVK_PC_STASH=64 -DVK_PC_TABLE_LOG2=4: k_path_count_agg with a stash of one round and a
# table of 16 entries, so that most crossings take the straight-to-memory arms).  Builds in a scratch copy and points the
# suite at it through VELLO_EMU_LIBRARY (tests/emu_lib.py): the tree's own libvello_emu.so is never touched, and a -D
# combination that does not compile stops the script (the build log is $SCRATCH/build.log).
#      EXTRA="-DVK_PC_STASH=64" bash scripts/emu_variant_check.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SCRATCH="${TMPDIR:-/tmp}/vello_emu_variant"
rm -rf "$SCRATCH" && mkdir -p "$SCRATCH/tests"
cp -r "$ROOT/tests/simt_emu" "$SCRATCH/tests/" && cp -r "$ROOT/vello_amd" "$ROOT/include" "$SCRATCH/"
cd "$SCRATCH/tests/simt_emu" && rm -rf build libvello_emu.so
if ! make -j8 CXXFLAGS="-O2 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mfma -I . -x c++ -Wno-unknown-pragmas ${EXTRA:-}" > "$SCRATCH/build.log" 2>&1; then
    tail -40 "$SCRATCH/build.log"
    echo "emu_variant_check: the variant build failed (EXTRA='${EXTRA:-}')" >&2
    exit 1
fi
test -f libvello_emu.so
cd "$ROOT"
VELLO_EMU_LIBRARY="$SCRATCH/tests/simt_emu/libvello_emu.so" python -m pytest tests/test_emu_parity.py -x -q -p no:cacheprovider "$@"
