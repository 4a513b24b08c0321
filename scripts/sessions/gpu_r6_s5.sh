#!/bin/bash
# Round 6 session 5: contexts of ONE build differ by 13 % in frames/s with four frames in flight -- stream -> hardware queue mapping?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s5
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(echo "== 8 contexts, no dummy streams"; timeout 300 python scripts/stream_order_probe.py 8 2>/dev/null
 echo "== 8 contexts, 1 dummy stream before each"; timeout 300 python scripts/stream_order_probe.py 8 1,1,1,1,1,1,1,1 2>/dev/null
 echo "== 4 contexts, 3 / 3 / 3 / 3 dummy streams"; timeout 300 python scripts/stream_order_probe.py 4 3,3,3,3 2>/dev/null
 echo "== GPU_MAX_HW_QUEUES=4, 6 contexts"; GPU_MAX_HW_QUEUES=4 timeout 300 python scripts/stream_order_probe.py 6 2>/dev/null
 echo "== GPU_MAX_HW_QUEUES=16, 6 contexts"; GPU_MAX_HW_QUEUES=16 timeout 300 python scripts/stream_order_probe.py 6 2>/dev/null
) > $OUT/stream_order_probe.txt
cat $OUT/stream_order_probe.txt
