#!/bin/bash
# same-box A/B of one stage under build variants (scripts/stage_ab.py + stage_times.py):  STAGE=path_count VARIANTS="H" bash scripts/gpu_stage_ab.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for v in A ${VARIANTS}; do timeout 200 python scripts/stage_ab.py $v ${STAGE} 2>/dev/null | tail -1 | tee -a gpurun_out/stage_ab.txt; done; done
for v in A ${VARIANTS}; do timeout 200 python scripts/stage_times.py $v 2>/dev/null | sed "s/^/$v /" | tee -a gpurun_out/stage_ab.txt; done
