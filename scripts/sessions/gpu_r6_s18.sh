#!/bin/bash
# Round 6 session 18: the tree (fine's slicing threshold 192 with frames in flight, binning + tile_alloc one launch) against HEAD 5f3.. (O), a process each;
# the whole GPU suite; the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s18
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4) > $OUT/gputest.log; tail -2 $OUT/gputest.log
timeout 120 python scripts/ab_process.py A d2 2>/dev/null > /dev/null
for rep in 1 2 3; do for L in O A; do timeout 120 python scripts/ab_process.py $L d2 r1mix mmark tiger 2>/dev/null; done; done > $OUT/ab_slices_binning.txt
cat $OUT/ab_slices_binning.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_k20.json; head -c 300 $OUT/bench_k20.json; echo
