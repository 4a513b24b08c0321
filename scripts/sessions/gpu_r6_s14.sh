#!/bin/bash
# Round 6 session 14: k_path_count's two footprints in the tree (one frame at a time: 48.7 KB; frames in flight: 24.6 KB): the GPU tests that
# walk it, the tree against itself three times (a process each), frames in flight 1-8 with 8 and 16 hardware queues, the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s14
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 900 python -m pytest tests -m gpu -q -x -k "path_count or c3_d2 or c5_all or front_fusion or pipelines" 2>&1 | tail -4) > $OUT/gputest_subset.log; tail -2 $OUT/gputest_subset.log
for rep in 1 2; do timeout 120 python scripts/ab_process.py A d2 r1mix 2>/dev/null; done > $OUT/ab_tree.txt; cat $OUT/ab_tree.txt
(GPU_MAX_HW_QUEUES=8 timeout 200 python scripts/inflight_probe.py d2 2>/dev/null; GPU_MAX_HW_QUEUES=16 timeout 200 python scripts/inflight_probe.py d2 2>/dev/null) > $OUT/inflight_probe.txt; cat $OUT/inflight_probe.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_k20.json; head -c 300 $OUT/bench_k20.json; echo
