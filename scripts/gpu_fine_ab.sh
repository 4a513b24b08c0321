#!/bin/bash
# same-box A/B of k_fine build variants (scripts/fine_ab.py):  VARIANTS="V2 V3" bash scripts/gpu_fine_ab.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for v in A ${VARIANTS}; do timeout 120 python scripts/fine_ab.py $v 2>/dev/null | tail -1 | tee -a gpurun_out/fine_ab.txt; done; done
