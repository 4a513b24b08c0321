#!/bin/bash
# Round 6 session 25: k_path_count's flush with the returning adds of 4 (tree) / 1 (O: as before) / 8 (E) turns in flight together
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s25
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "path_count or config_c3 or back_half or c5" 2>&1 | tail -3) > $OUT/gputest.log; cat $OUT/gputest.log
for rep in 1 2 3; do for L in O A E; do timeout 120 python scripts/ab_process.py $L d2 r1mix mmark 2>/dev/null; done; done > $OUT/ab_path_count_flush_group.txt
cat $OUT/ab_path_count_flush_group.txt
