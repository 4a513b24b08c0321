#!/bin/bash
# round 5, session 15: k_path_count with the next chunk's lines requested a chunk ahead (A) against the same tree without (N);
# coarse's and fine's phase profiles on d2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s15
mkdir -p $O
rm -f gpurun_out/stage_ab.txt
STAGE=path_count VARIANTS="N" bash scripts/gpu_stage_ab.sh 2>&1 | grep -v amdgpu.ids | tee $O/path_count_ab.txt
timeout 300 python scripts/coarse_prof.py 2>&1 | grep -v amdgpu.ids | tee $O/coarse_prof.txt
timeout 300 python scripts/fine_prof.py d2 2>&1 | grep -v amdgpu.ids | tee $O/fine_prof.txt
