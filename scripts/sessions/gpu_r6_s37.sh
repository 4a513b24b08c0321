#!/bin/bash
# Round 6 session 37: one frame at a time, the stroke workgroups of k_flatten_main as a grid that strides (K 512, M 768, L 1024 workgroups; tree 4096 = a round each):
# with the tag words a round ahead a workgroup that walks several rounds starts each with its loads on the way
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s37
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in A K M L; do timeout 120 python scripts/ab_process.py $L d2 r1mix 2>/dev/null | cut -c1-150; done; done > $OUT/ab_strokes_grid_alone.txt
cat $OUT/ab_strokes_grid_alone.txt
