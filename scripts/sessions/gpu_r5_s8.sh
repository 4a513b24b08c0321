#!/bin/bash
# round 5, session 8: A = tree (dense lists without stroked curves through flatten_tag again), R4; W = rare_command inlined into the
# brush kernels (no RareState round trip through scratch): the brush workloads' stages, d2's fine
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s8
mkdir -p $O
for v in A R4 A R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
for v in A W A W; do VELLO_AB_LIB=$([ $v = A ] && echo "" || echo $v) timeout 200 python scripts/brush_prof.py stages 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $O/brush_stages.txt; done
VARIANTS="W" REPS="1 2" STEPS=100 bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
