#!/bin/bash
# round 5, session 9: the tree against R4 -- the whole GPU suite, flatten's kernels, the bench A/B, the other workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s9
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/gputest.txt
for v in A R4 A R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
VARIANTS="R4" REPS="1 2" STEPS=100 bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
timeout 300 python scripts/other_workloads.py 2>/dev/null | tee $O/other_workloads.jsonl
VELLO_AB_LIB=R4 timeout 300 python scripts/other_workloads.py 2>/dev/null | tee $O/other_workloads_R4.jsonl
