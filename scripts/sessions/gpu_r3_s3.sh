#!/bin/bash
# Round 3, session 3: GPU suite at the working tree (fine: slot parameters in lane registers, record preload, sliced long tiles),
# then same-box A/B: S = previous commit, A = in tree, N = slices never cut (their empty blocks still launched), Z = no slice blocks.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s3
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $OUT/gputest.log; tail -5 $OUT/gputest.log
for rep in 1 2; do
  VARIANTS="${VARIANTS:-S N Z}" REPS=1 bash scripts/gpu_ab.sh | tee -a $OUT/ab.txt
done
