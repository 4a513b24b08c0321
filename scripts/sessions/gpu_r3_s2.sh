#!/bin/bash
# Round 3, session 2: GPU suite at the working tree (DPP scans, marker-based item pass, regular-prefix replay loop), k_fine phase
# timers, same-box A/B: B = round 2 kernels + merged restore, A = in tree, S / M = fine.hip with -sink-insts-to-avoid-spills /
# -disable-machine-licm, K = coarse.hip with -sink-insts-to-avoid-spills.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s2
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout 90 python scripts/fine_prof.py d2 r1mix > $OUT/fine_prof.txt 2>&1; tail -44 $OUT/fine_prof.txt
for rep in 1 2; do
  VARIANTS="${VARIANTS:-B S M K}" REPS=1 bash scripts/gpu_ab.sh | tee -a $OUT/ab.txt
done
