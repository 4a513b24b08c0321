#!/bin/bash
# Memory-safety sweep on the CPU: the kernel sources (SIMT-emulator build) and the host code compiled with
# AddressSanitizer, then the emulator parity suite -- tiger, brushes, catalogue scenes, the whole-API fuzzers, the
# malformed-scene cases -- run against that library.  Device buffers are heap blocks in the emulator, so an index a
# kernel computes one element out of bounds is caught here instead of being a silent wild read on the GPU (HIP has no
# robust buffer access).  Leaves the repository untouched: builds in a scratch copy, which the suites are pointed at.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SCRATCH="${TMPDIR:-/tmp}/vello_emu_asan"
rm -rf "$SCRATCH" && mkdir -p "$SCRATCH/tests"
cp -r "$ROOT/tests/simt_emu" "$SCRATCH/tests/" && cp -r "$ROOT/vello_amd" "$ROOT/include" "$SCRATCH/"
cd "$SCRATCH/tests/simt_emu" && rm -rf build libvello_emu.so
make -j8 CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mfma -I . -x c++ -fsanitize=address -fno-omit-frame-pointer -Wno-unknown-pragmas" >/dev/null 2>&1 || true
g++ -shared -fsanitize=address -o libvello_emu.so build/*.o
cd "$ROOT"
# (the suites load the library VELLO_EMU_LIBRARY names -- tests/emu_lib.py -- instead of the in-tree build)
VELLO_EMU_LIBRARY="$SCRATCH/tests/simt_emu/libvello_emu.so" ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so)" \
    python -m pytest tests/test_emu_parity.py tests/test_distributed_gloo.py -x -q -p no:cacheprovider "$@"
