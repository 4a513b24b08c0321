#!/bin/bash
# Round 6 session 1: the round's starting point on this round's boxes -- the GPU suite (with the new C5 d2-seed test) and the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s1
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_k20.json; head -c 400 $OUT/bench_k20.json; echo
