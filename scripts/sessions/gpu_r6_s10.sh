#!/bin/bash
# Round 6 session 10: ring first = +8 %: torch's stream used before the engine's (queues) or the allocation order (placement)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s10
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(for m in tiny_first empty_first ring_first ring_last tiny_first empty_first; do timeout 200 python scripts/bench_loop_probe.py $m 2>/dev/null | grep '"bare"'; done
) > $OUT/alloc_order_probe2.txt
cat $OUT/alloc_order_probe2.txt
