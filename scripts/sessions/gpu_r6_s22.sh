#!/bin/bash
# Round 6 session 22: k_fine's LDS: 512 records per batch (9.4 KB: 17 waves per CU at most) against 256 (V1) / 320 (V2) (8.4 / 8.6 KB: 19 waves per CU with the five-wave form)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s22
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 120 python scripts/ab_process.py A d2 2>/dev/null > /dev/null
for rep in 1 2 3; do for L in A V1 V2; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null; done; done > $OUT/ab_fine_item_cap.txt
cat $OUT/ab_fine_item_cap.txt
