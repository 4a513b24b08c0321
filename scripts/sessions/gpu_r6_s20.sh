#!/bin/bash
# Round 6 session 20: is the GPU fed fast enough (host enqueue cost, two host threads with a context each), and the run loop of k_fine (A) against HEAD (O)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s20
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 120 python scripts/ab_process.py A d2 2>/dev/null > /dev/null
(timeout 200 python scripts/host_feed_probe.py 2 2>/dev/null; timeout 200 python scripts/host_feed_probe.py 3 2>/dev/null) > $OUT/host_feed_probe.txt; cat $OUT/host_feed_probe.txt
for rep in 1 2 3; do for L in O A; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null; done; done > $OUT/ab_fine_run_loop.txt
cat $OUT/ab_fine_run_loop.txt
