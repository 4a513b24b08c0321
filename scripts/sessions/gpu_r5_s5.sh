#!/bin/bash
# round 5, session 5: A = tree (cooperative flattener with the list density by composition + ms_fill_simple) against R4:
# bench A/B with k_fine's SQ instruction counters, flatten's kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s5
mkdir -p $O
for v in A R4 A R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
VARIANTS="R4" PMC=1 REPS="1 2" STEPS=100 bash scripts/gpu_r4_ab.sh 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
cp -r gpurun_out/r4ab $O/
