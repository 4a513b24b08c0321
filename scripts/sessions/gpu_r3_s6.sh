#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s6
mkdir -p $OUT
(timeout 300 python -m pytest tests -m gpu -q -x -k "slices or d2_scene or tiger or mmark_50k" 2>&1 | tail -15) > $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout 120 python scripts/fine_timeline.py d2 > $OUT/timeline_slices_d2.txt 2>&1; head -8 $OUT/timeline_slices_d2.txt; tail -8 $OUT/timeline_slices_d2.txt
for rep in 1 2 3; do
  VARIANTS="${VARIANTS:-Z}" REPS=1 bash scripts/gpu_ab.sh | tee -a $OUT/ab.txt
done
