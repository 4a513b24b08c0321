#!/bin/bash
# Round 6 session 41: the whole library built with one extra LLVM AMDGPU option each (scripts/variant_flags.sh): P -amdgpu-use-amdgpu-trackers=1,
# Q -amdgpu-early-ifcvt=1, R -amdgpu-sched-strategy=max-ilp, S -amdgpu-schedule-metric-bias=0, against HEAD's library (H); a process each, alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s41
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in H P Q R S; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-270; done; done > $OUT/ab_llvm_options.txt
cat $OUT/ab_llvm_options.txt
