#!/bin/bash
# round 5, session 25: the zero fill of a frame's tiles in k_pathtag_scan's launch / in k_flatten_light's / in tile_alloc
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s25
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
(timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "tiles_zeroed" 2>&1 | tail -8) > $O/new_tests.log; tail -3 $O/new_tests.log
timeout 300 python scripts/experiments/round5b_ab3.py 3 > $O/ab3.jsonl 2> $O/ab3.txt; grep -v amdgpu.ids $O/ab3.txt
