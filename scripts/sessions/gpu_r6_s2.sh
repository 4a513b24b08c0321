#!/bin/bash
# Round 6 session 2: k_path_count's LDS footprint, compile-time constants only, contexts of one process:
# A = the tree (table 512 entries, stash 768: 48.7 KB), P = table 256 + stash 512 (26.5 KB), Q = P on a grid of 1024, R = table 256 + stash 384 + chunks of 512 lines on 1024
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s2
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 500 python scripts/ab_contexts.py APQR 3 d2 r1mix > $OUT/ab_path_count_footprint.jsonl 2> $OUT/ab_path_count_footprint.txt
cat $OUT/ab_path_count_footprint.txt
