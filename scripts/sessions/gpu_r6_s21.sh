#!/bin/bash
# Round 6 session 21: k_fine capped at 96 VGPRs (W5: launch bounds of five waves per SIMD; the LDS still admits 17 waves per CU) -- do the other frames'
# bandwidth-bound kernels run in the registers it leaves free?  A = the tree (128 VGPRs: four waves per SIMD are the whole register file)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s21
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 120 python scripts/ab_process.py A d2 2>/dev/null > /dev/null
for rep in 1 2 3; do for L in A W5; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null; done; done > $OUT/ab_fine_vgpr_cap.txt
cat $OUT/ab_fine_vgpr_cap.txt
