#!/bin/bash
# Round 6 session 28: k_fine with a simple fill's touched pixels from their records in registers (VK_FINE_REGS=1: the tree's library in this
# session, A) against HEAD (H); the GPU suite on the variant
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s28
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3) > $OUT/gputest.log; cat $OUT/gputest.log
for rep in 1 2 3; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 mmark r1mix 2>/dev/null; done; done > $OUT/ab_fine_regs.txt
cat $OUT/ab_fine_regs.txt
