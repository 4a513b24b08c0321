#!/bin/bash
# round 5, session 18: k_coarse with two list entries per thread and stream round (A) against the commit before it (P): the stage,
# the back half of the GPU suite, coarse's phase profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s18
mkdir -p $O
rm -f gpurun_out/stage_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c3 or d2 or mmark or tiger or clip or blend or many or random or smoke or brush or gradient" 2>&1 | tail -3 | tee $O/tests.txt
STAGE=coarse VARIANTS="P" bash scripts/gpu_stage_ab.sh 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $O/coarse_ab.txt
timeout 300 python scripts/coarse_prof.py d2 2>&1 | grep -v amdgpu.ids | tee $O/coarse_prof.txt
