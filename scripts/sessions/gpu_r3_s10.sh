#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s10
mkdir -p $OUT
(timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $OUT/gputest.log; tail -3 $OUT/gputest.log
for rep in 1 2; do
  VARIANTS="${VARIANTS:-H}" REPS=1 bash scripts/gpu_ab.sh | tee -a $OUT/ab.txt
done
