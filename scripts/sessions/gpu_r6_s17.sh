#!/bin/bash
# Round 6 session 17: fine's slicing threshold with frames in flight (A = 128 in the tree; S 96 = before, T 160, U 192, V 256, W never; X = 128 in slices of 48, Y = 160 in slices of 64)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s17
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 120 python scripts/ab_process.py A d2 2>/dev/null > /dev/null
for rep in 1 2; do for L in A S T U V W X Y; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null; done; done > $OUT/ab_slices_in_flight.txt
cat $OUT/ab_slices_in_flight.txt
