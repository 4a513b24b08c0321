#!/bin/bash
# round 5, session 22: from which crossing on a long line's crossings are spread over the lanes (k_path_count, VK_PC_COOP_FROM):
# 8 (A), 12, 16 against the commit before (P)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s22
mkdir -p $O
rm -f gpurun_out/stage_ab.txt
for rep in 1 2; do for v in A K12 K16 P; do timeout 300 python scripts/stage_small.py $v path_count 2>/dev/null | tail -1; done; done | tee $O/stage_small.txt
for rep in 1 2 3; do for v in A K12 K16 P; do timeout 200 python scripts/stage_ab.py $v path_count 2>/dev/null | tail -1; done; done | tee $O/stage_ab.txt
