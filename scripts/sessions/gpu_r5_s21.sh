#!/bin/bash
# round 5, session 21: k_path_count's counting pass with the long lines of a wave spread over its lanes (A) against the commit before
# (P = b83065c): the GPU tests that walk it, the stage on d2 / r1mix and on the smaller workloads, the tiger's chunk time line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s21
mkdir -p $O
rm -f gpurun_out/stage_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "path_count or tiger or d2 or mmark or c3 or smoke or random or catalogue or long_lines or fusion" 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2; do for v in A P; do timeout 300 python scripts/stage_small.py $v path_count 2>/dev/null | tail -1; done; done | tee $O/stage_small.txt
STAGE=path_count VARIANTS="P" bash scripts/gpu_stage_ab.sh 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $O/stage_ab.txt
timeout 300 python scripts/pc_timeline.py tiger 2>&1 | grep -v amdgpu.ids | head -8 | tee $O/pc_timeline_tiger.txt
