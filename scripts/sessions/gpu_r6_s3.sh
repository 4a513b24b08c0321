#!/bin/bash
# Round 6 session 3: k_path_count's footprint, second sweep (see scripts/sessions/gpu_r6_s2.sh): LPT 2 / 1 with smaller tables and stashes, grids of 3-8 workgroups per CU
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s3
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 800 python scripts/ab_contexts.py ARBCDEFGH 3 d2 r1mix > $OUT/ab_path_count_footprint2.jsonl 2> $OUT/ab_path_count_footprint2.txt
cat $OUT/ab_path_count_footprint2.txt
