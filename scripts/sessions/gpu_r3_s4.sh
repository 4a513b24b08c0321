#!/bin/bash
# Round 3, session 4: sliced fine with the write-through hand-off and 8 coverage loads in flight: sliced GPU tests, then A/B
# against Z (no slices).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s4
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 300 python -m pytest tests -m gpu -q -x -k "slices or d2_scene or tiger or mmark_50k" 2>&1 | tail -15) > $OUT/gputest.log; tail -3 $OUT/gputest.log
for rep in 1 2 3; do
  VARIANTS="${VARIANTS:-Z}" REPS=1 bash scripts/gpu_ab.sh | tee -a $OUT/ab.txt
done
