#!/bin/bash
# Round 6 session 23: k_backdrop loads only the 128-byte lines of the pool that k_path_count flagged (a backdrop bump landed there): the GPU suite, A (tree) against O (HEAD)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s23
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4) > $OUT/gputest.log; tail -2 $OUT/gputest.log
for rep in 1 2 3; do for L in O A; do timeout 120 python scripts/ab_process.py $L d2 r1mix mmark tiger 2>/dev/null; done; done > $OUT/ab_backdrop_flags.txt
cat $OUT/ab_backdrop_flags.txt
