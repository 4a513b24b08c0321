#!/bin/bash
# Round 6 session 26: (H) HEAD 0cc4c98 against the tree (A: ms_fill_simple's addend as bit operation + xor-add); k_path_count's in-flight form again behind the
# flush change: (P) the one-frame form in flight too, (Q) the in-flight form with a table of 512 lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s26
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in H A P Q; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null; done; done > $OUT/ab_fine_addend_pc_forms.txt
cat $OUT/ab_fine_addend_pc_forms.txt
