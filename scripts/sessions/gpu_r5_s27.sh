#!/bin/bash
# round 5, session 27: coarse's long lists split over two workgroups per quadrant (A) against the commit before (Q); the forced-split tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s27
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "coarse_split or config_c3 or config_c4 or config_c2" 2>&1 | tail -4) > $O/new_tests.log; tail -3 $O/new_tests.log
timeout 300 python scripts/ab_contexts.py AQAQ 2 > $O/ab.jsonl 2> $O/ab.txt; grep -v amdgpu.ids $O/ab.txt
