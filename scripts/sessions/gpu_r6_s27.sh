#!/bin/bash
# Round 6 session 27: k_path_count's counting pass with the line groups of a wave side by side (tree A) against HEAD f93a2cc (H); parity; the chunk time line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s27
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -k "path_count or config_c or back_half or tiger or fuzz" 2>&1 | tail -3) > $OUT/gputest.log; cat $OUT/gputest.log
for rep in 1 2 3; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 r1mix mmark 2>/dev/null; done; done > $OUT/ab_path_count_groups.txt
cat $OUT/ab_path_count_groups.txt
python scripts/pc_timeline.py d2 2>&1 | grep -v amdgpu.ids | head -8 > $OUT/pc_timeline_d2_groups.txt; cat $OUT/pc_timeline_d2_groups.txt
