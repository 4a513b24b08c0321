#!/bin/bash
# Round 4 evidence for one commit, one GPU session: the GPU suite, the bench line as the driver runs it (--steps 20 --warmup 5) and
# with the defaults (CPU baseline included), rocprofv3 kernel stats / PMC traffic / SQ counters (scripts/gpu_profiles.sh), the
# kernel time line with frames in flight, fine's phase profile if a measurement build travels along.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r4_evidence
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/r04_gputest.log; tail -2 $OUT/r04_gputest.log
timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/r04_bench_driver_flags_k20.json; head -c 300 $OUT/r04_bench_driver_flags_k20.json; echo
timeout 400 python bench.py 2>/dev/null | tail -1 > $OUT/r04_bench.json; head -c 300 $OUT/r04_bench.json; echo
TAG=r04 bash scripts/gpu_profiles.sh > $OUT/profiles.log 2>&1; tail -3 $OUT/profiles.log
NIF="4 1" bash scripts/gpu_r4_timeline.sh > $OUT/r04_pipeline_timeline.txt 2>&1; grep -A3 "window" $OUT/r04_pipeline_timeline.txt | head -12
timeout 120 scripts/calib/valu_rate > $OUT/r04_valu_rate.txt 2>&1
