#!/bin/bash
# round 5, session 2: the wave-cooperative Euler flattener (A = tree; R4 = round 4's library; B / C = other list packings),
# parity of everything flatten touches, its phase profile, and what session 1 got wrong (brush phases with a stale PROF library,
# the 4 x scene without a bin_data pool).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -x -q -k "primitive or tiger or mmark or stroke or cardioid or tricky or d2 or funky or robust or fuzz or flatten or c3 or random" 2>&1 | tail -6 | tee $O/tests.txt
for v in A R4 B C A R4; do timeout 200 python scripts/stage_times.py $v 2>/dev/null | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $O/stage_times.txt; done
timeout 300 python scripts/flatten_prof.py mmark tiger d2 2>&1 | grep -v amdgpu.ids | tee $O/flatten_prof.txt
timeout 200 python scripts/brush_prof.py phases 2>&1 | grep -v amdgpu.ids | tee $O/brush_phases.txt
timeout 400 python scripts/batch_estimate.py 2>&1 | grep -v amdgpu.ids | tee $O/batch_estimate.txt
