#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s8
mkdir -p $OUT
(timeout 300 python -m pytest tests -m gpu -q -x -k "slices or d2_scene or tiger or mmark_50k or paris" 2>&1 | tail -15) > $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout 120 python scripts/fine_timeline.py d2 r1mix > $OUT/timeline.txt 2>&1; grep -E "waves logged|tile  |slice|late wave" $OUT/timeline.txt | head -40
for rep in 1 2; do
  VARIANTS="${VARIANTS:-H T128 T80 F48}" REPS=1 bash scripts/gpu_ab.sh | tee -a $OUT/ab.txt
done
