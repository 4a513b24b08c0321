#!/bin/bash
# Round 6 session 30: smaller grids for the in-flight forms of the stroke kernel and k_path_count together (tree 512 / 1024): K 384 / 768, L 256 / 768, M 384 / 512,
# N 256 / 512, R 320 / 640.  A process per build, alternating; d2 and mmark.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s30
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in A K L M N R; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-60; done; done > $OUT/ab_in_flight_grids.txt
cat $OUT/ab_in_flight_grids.txt
