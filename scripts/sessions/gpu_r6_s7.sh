#!/bin/bash
# Round 6 session 7: what in bench.py's timed loop costs 8 % against a bare loop (events around the dominant kernel? the pipeline's waits?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s7
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 300 python scripts/bench_loop_probe.py 2>/dev/null > $OUT/bench_loop_probe.txt
cat $OUT/bench_loop_probe.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | head -c 300; echo
