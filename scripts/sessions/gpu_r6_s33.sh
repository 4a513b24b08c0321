#!/bin/bash
# Round 6 session 33: wave_bbox_update leaves at once when no lane of the wave has an extent (tree A) against HEAD (H)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s33
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "stroke or config_c3 or flatten or c5" 2>&1 | tail -2) > $OUT/gputest.log; cat $OUT/gputest.log
for rep in 1 2 3 4; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-140; done; done > $OUT/ab_bbox_early_out.txt
cat $OUT/ab_bbox_early_out.txt
