#!/bin/bash
# Round 6 session 34: k_backdrop requesting nothing beyond its block's last tile (K: VK_BD_CLAMP_TO_BLOCK) against the tree (A) -- round 4 kept the requests beyond
# the block for 3 us one frame at a time; with frames in flight the traffic may count for more
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s34
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3 4; do for L in A K; do timeout 120 python scripts/ab_process.py $L d2 2>/dev/null | cut -c1-260; done; done > $OUT/ab_backdrop_clamp.txt
cat $OUT/ab_backdrop_clamp.txt
