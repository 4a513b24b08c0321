#!/bin/bash
# round 5, session 26: how the zero fill in k_pathtag_scan's launch makes its stores (what stays dirty in L2 is flushed at the launch's end)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s26
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
for m in 1 2 3; do (VELLO_HIP_PREZERO_MODE=$m timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -k "tiles_zeroed" 2>&1 | tail -2) ; done > $O/new_tests.log; cat $O/new_tests.log
timeout 300 python scripts/experiments/round5b_ab4.py 3 > $O/ab4.jsonl 2> $O/ab4.txt; grep -v amdgpu.ids $O/ab4.txt
