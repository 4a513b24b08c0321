#!/bin/bash
# round 5, session 24: the library or the order of the contexts?  (follow-up to s23: d2's unchanged kernels were 4-11 % slower in the
# contexts created second and third); the pre-zero test with its GPU form of the pool comparison
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s24
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
(timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "tiles_zeroed" 2>&1 | tail -8) > $O/new_tests.log; tail -3 $O/new_tests.log
timeout 200 python scripts/experiments/round5b_ab2.py APAP 3 > $O/ab2_APAP.jsonl 2> $O/ab2_APAP.txt; grep -v amdgpu.ids $O/ab2_APAP.txt
timeout 200 python scripts/experiments/round5b_ab2.py PAPA 3 > $O/ab2_PAPA.jsonl 2> $O/ab2_PAPA.txt; grep -v amdgpu.ids $O/ab2_PAPA.txt
