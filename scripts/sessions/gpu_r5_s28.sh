#!/bin/bash
# round 5, session 28: tag words per thread of the pathtag scan -- 8 (A, the tree) against 4 (F, as it was) and 16 (S)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s28
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
timeout 300 python scripts/ab_contexts.py AFSAFS 2 > $O/ab.jsonl 2> $O/ab.txt; grep -v amdgpu.ids $O/ab.txt | cut -c1-330
