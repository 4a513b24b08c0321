#!/bin/bash
# Round 6 session 13: k_path_count's footprint again, a PROCESS per build (its context the only one: the same hardware queues every time), alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s13
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in A R H B P D E; do timeout 120 python scripts/ab_process.py $L d2 r1mix 2>/dev/null; done; done > $OUT/ab_path_count_footprint_processes.txt
cat $OUT/ab_path_count_footprint_processes.txt
