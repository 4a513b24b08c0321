#!/bin/bash
# round 5, session 7: where do k_flatten_main's +11 / +18 us on d2 / r1mix come from?  V1 = the tree with every lane on its own
# (no wave ever walks together): the new structure without the cooperation.  + the flatten phase profile of r1mix.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s7
mkdir -p $O
for v in A V1 R4 A V1 R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
timeout 300 python scripts/flatten_prof.py r1mix d2 2>&1 | grep -v amdgpu.ids | tee $O/flatten_prof.txt
