#!/bin/bash
# round 5, session 13: two sets of kernels for flatten's heavy list, picked by the engine from what a finished frame of the scene
# put on the list: flatten's kernels against R4, the new GPU tests, the bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s13
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "flatten_kernel_sets" 2>&1 | tail -3 | tee $O/tests.txt
for v in A R4 A R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
VARIANTS="R4" REPS="1 2" STEPS=100 bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
