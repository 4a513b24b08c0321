#!/bin/bash
# Round 6 session 9: one context, 2125 frames/s in bench_loop_probe against 2300 as stream_order_probe's first context: the order of allocations?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s9
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(for m in ring_first ring_mid ring_last ring_first; do timeout 200 python scripts/bench_loop_probe.py $m 2>/dev/null | grep '"bare"'; done
 echo "== stream_order_probe 2"; timeout 300 python scripts/stream_order_probe.py 2 2>/dev/null
) > $OUT/alloc_order_probe.txt
cat $OUT/alloc_order_probe.txt
