#!/bin/bash
# Round 6 session 24: k_coarse with four waves per workgroup (46 KB of LDS instead of 86: VK_COARSE_NW=4), a process per build, alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s24
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in A C; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null; done; done > $OUT/ab_coarse_nw4.txt
cat $OUT/ab_coarse_nw4.txt
