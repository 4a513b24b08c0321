#!/bin/bash
# round 5, session 11: staging back at 3072 lines (EC_PIECES 16): flatten's kernels against R4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s11
mkdir -p $O
for v in A R4 A R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
