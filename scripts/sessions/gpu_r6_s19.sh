#!/bin/bash
# Round 6 session 19: the kernel time line with four frames in flight on the tree (queue order, k_path_count's in-flight form, no slices in flight)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s19
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
NIF="4 1" timeout 300 bash scripts/gpu_r4_timeline.sh > $OUT/r06_pipeline_timeline_gaps.txt 2>&1
cat $OUT/r06_pipeline_timeline_gaps.txt
