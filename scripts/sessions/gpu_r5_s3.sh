#!/bin/bash
# round 5, session 3: the cooperative flattener with its ranges in LDS and the fp64 routines out of line (no spills);
# A = tree, R4 = round 4's library, B / C = the heavy list over 2048 / 3072 waves.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s3
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_primitives.py -x -q 2>&1 | grep -E "differ|rror|passed|failed" | head -5 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tiger or mmark or stroke or cardioid or tricky or d2 or funky or robust or fuzz or flatten or c3 or random" 2>&1 | tail -4 | tee -a $O/tests.txt
for v in A R4 B C A R4; do timeout 200 python scripts/stage_times.py $v 2>/dev/null | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $O/stage_times.txt; done
timeout 300 python scripts/flatten_prof.py mmark tiger d2 2>&1 | grep -v amdgpu.ids | tee $O/flatten_prof.txt
