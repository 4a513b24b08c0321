#!/bin/bash
# Round 6 session 8: the lanes' streams with a priority of their own (a hardware-queue pool per priority): are contexts then equal whatever was created before them?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s8
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(echo "== default priority, 6 contexts"; timeout 300 python scripts/stream_order_probe.py 6 2>/dev/null
 echo "== VELLO_HIP_LANE_PRIORITY=high, 6 contexts"; VELLO_HIP_LANE_PRIORITY=high timeout 300 python scripts/stream_order_probe.py 6 2>/dev/null
 echo "== VELLO_HIP_LANE_PRIORITY=low, 6 contexts"; VELLO_HIP_LANE_PRIORITY=low timeout 300 python scripts/stream_order_probe.py 6 2>/dev/null
 echo "== bench loop probe, default"; timeout 300 python scripts/bench_loop_probe.py 2>/dev/null | grep -v "all"
 echo "== bench loop probe, high"; VELLO_HIP_LANE_PRIORITY=high timeout 300 python scripts/bench_loop_probe.py 2>/dev/null | grep -v "all"
) > $OUT/lane_priority_probe.txt
cat $OUT/lane_priority_probe.txt
