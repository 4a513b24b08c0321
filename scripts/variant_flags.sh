#!/bin/bash
# Builds the in-tree kernel sources with extra compile-time constants without touching the product:
#   scripts/variant_flags.sh <LETTER> "-DVK_PC_TABLE_LOG2=8 -DVK_PC_STASH=512" [patch-name]
# -> ab_tmp/libvello_hip_<LETTER>.so (git-ignored, travels with gpurun) for scripts/ab_contexts.py.  An optional
# scripts/experiments/<patch-name>.patch is applied first.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LETTER=$1; FLAGS=$2; PATCH=${3:-}
W="${TMPDIR:-/tmp}/vello_variant_$LETTER"
rm -rf "$W" && mkdir -p "$W/vello_amd" "$ROOT/ab_tmp"
cp -r "$ROOT/vello_amd/csrc" "$W/vello_amd/" && cp -r "$ROOT/include" "$W/"
rm -rf "$W/vello_amd/csrc/build"
if [ -n "$PATCH" ]; then (cd "$W" && patch -p1 --no-backup-if-mismatch < "$ROOT/scripts/experiments/$PATCH.patch"); fi
make -s -j8 -C "$W/vello_amd/csrc" EXTRA="$FLAGS" 2>&1 | grep -E "error|Error" || true
cp "$W/vello_amd/lib/libvello_hip.so" "$ROOT/ab_tmp/libvello_hip_$LETTER.so"
echo "built ab_tmp/libvello_hip_$LETTER.so with $FLAGS $PATCH"
