#!/bin/bash
# round 5, session 4: flatten's kernels one by one (A = tree with the list over 3072 waves, R4, D = 4096, E = 6144), the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s4
mkdir -p $O
for v in A R4 D E A R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/gputest.txt
