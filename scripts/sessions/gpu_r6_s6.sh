#!/bin/bash
# Round 6 session 6: which hardware queues do the lanes of the probe's contexts dispatch on (rocprofv3 kernel trace: Queue_Id per Stream_Id)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s6
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_q -o p -- python scripts/stream_order_probe.py 4 > $OUT/probe4.txt 2> $OUT/probe4.err
f=$(find $OUT/tmp_q -name "*kernel_trace.csv" | head -1)
(cat $OUT/probe4.txt; python scripts/queue_map.py $f) > $OUT/queue_map_probe4.txt
rm -rf $OUT/tmp_q
cat $OUT/queue_map_probe4.txt
