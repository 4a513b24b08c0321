#!/bin/bash
# Round 6 session 15: same box, a process each, alternating: O = the tree before k_path_count's in-flight form (HEAD f7d..: first-queue fix in), A = the tree, R = the constants everywhere
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s15
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in O A R; do timeout 120 python scripts/ab_process.py $L d2 r1mix mmark 2>/dev/null; done; done > $OUT/ab_path_count_in_tree.txt
cat $OUT/ab_path_count_in_tree.txt
