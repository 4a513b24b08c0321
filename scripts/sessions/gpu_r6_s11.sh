#!/bin/bash
# Round 6 session 11: the hardware queues of the lanes with torch's stream used before / after the engine exists (rocprofv3 Queue_Id per Stream_Id)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s11
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for m in tiny_first ring_last; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tmp_$m -o p -- python scripts/bench_loop_probe.py $m > $OUT/probe_$m.txt 2> $OUT/probe_$m.err
  f=$(find $OUT/tmp_$m -name "*kernel_trace.csv" | head -1)
  (echo "== $m"; grep '"bare"' $OUT/probe_$m.txt; python scripts/queue_map.py $f) >> $OUT/queue_map_modes.txt
  rm -rf $OUT/tmp_$m
done
cat $OUT/queue_map_modes.txt
