#!/bin/bash
# round 5, session 12: A = the walk twice (inline math for lanes on their own, out-of-line math under the cooperative walk), A0 = the
# last commit (out-of-line math everywhere), R4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s12
mkdir -p $O
for v in ${VLIST:-A A0 R4 A A0 R4}; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
