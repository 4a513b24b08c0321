#!/bin/bash
# Round 6 session 4: is H's gain (table 256, stash 256, chunks of 512 lines, grid 1024) the build or the order the contexts were created in?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s4
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 800 python scripts/ab_contexts.py HARHAR 3 d2 r1mix > $OUT/ab_path_count_footprint3.jsonl 2> $OUT/ab_path_count_footprint3.txt
cat $OUT/ab_path_count_footprint3.txt
