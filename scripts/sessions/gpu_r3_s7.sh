#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s7
mkdir -p $OUT
(timeout 300 python -m pytest tests -m gpu -q -x -k "slices or d2_scene or tiger or mmark_50k" 2>&1 | tail -15) > $OUT/gputest.log; tail -3 $OUT/gputest.log
for rep in 1 2; do
  VARIANTS="${VARIANTS:-T64 T48 Z}" REPS=1 bash scripts/gpu_ab.sh | tee -a $OUT/ab.txt
done
