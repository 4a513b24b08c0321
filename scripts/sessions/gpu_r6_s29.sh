#!/bin/bash
# Round 6 session 29: launch constants of the in-flight forms again, behind the round's changes: the stroke kernel's grid (K 768, L 1024, M 384; tree 512),
# k_path_count's in-flight grid (N 768, R 1536; tree 1024), fine's slicing threshold with frames in flight (S 128; tree 192).  A process per build, alternating.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s29
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in A K L M N R S; do timeout 120 python scripts/ab_process.py $L d2 2>/dev/null | cut -c1-60; done; done > $OUT/ab_in_flight_constants.txt
cat $OUT/ab_in_flight_constants.txt
