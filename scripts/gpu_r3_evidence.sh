#!/bin/bash
# Round 3 evidence for one commit, one GPU session: the GPU suite, the bench line (default flags, CPU baseline included), rocprofv3
# kernel stats / PMC traffic / SQ counters (scripts/gpu_profiles.sh), the other workloads and their stage times.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3_evidence
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/r03_gputest.log; tail -2 $OUT/r03_gputest.log
timeout 400 python bench.py 2>/dev/null | tail -1 > $OUT/r03_bench.json; head -c 400 $OUT/r03_bench.json; echo
TAG=r03 bash scripts/gpu_profiles.sh > $OUT/profiles.log 2>&1; tail -3 $OUT/profiles.log
timeout 300 python scripts/other_workloads.py > $OUT/r03_other_workloads.jsonl 2>/dev/null; cat $OUT/r03_other_workloads.jsonl | cut -c1-160
timeout 200 python scripts/stage_times.py > $OUT/r03_stage_times.txt 2>/dev/null; cat $OUT/r03_stage_times.txt
