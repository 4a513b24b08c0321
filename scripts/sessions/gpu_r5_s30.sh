#!/bin/bash
# round 5, session 30: flatten's stroke workgroups with their list entries two rounds and their tag words + monoids one round ahead (A)
# against the tree at a79de23 (E); the stroke-kernel tests first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s30
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "stroke or config_c3 or config_c4 or config_c2 or tricky or kernel_sets" 2>&1 | tail -4) > $O/new_tests.log; tail -3 $O/new_tests.log
timeout 300 python scripts/ab_contexts.py AEAE 3 > $O/ab.jsonl 2> $O/ab.txt; grep -v amdgpu.ids $O/ab.txt | cut -c1-330
