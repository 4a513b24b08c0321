#!/bin/bash
# Round 6 session 16: other kernels' launch shapes under overlap (a process per build): S / T = k_flatten_strokes on 256 / 1024 workgroups (512),
# U / V = the heavy list over 512 / 2048 waves (1024), W / X = tiles sliced from 64 / 128 fills (96), Y = slices of 48 fills (32)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s16
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2; do for L in A S T U V W X Y; do timeout 120 python scripts/ab_process.py $L d2 2>/dev/null; done; done > $OUT/ab_launch_shapes.txt
cat $OUT/ab_launch_shapes.txt
