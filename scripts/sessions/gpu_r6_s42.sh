#!/bin/bash
# Round 6 session 42: early occlusion (path.hip: the fills' lines first, a table of every screen tile's last opaque full cover, the strokes' crossings under
# it dropped before their tile atomics / SegmentCount records / k_path_tiling threads) -- tree (A) against HEAD's library (H)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s42
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "early_occ or config_c or back_half or tiger or catalogue or fuzz or path_count" 2>&1 | tail -30 | cut -c1-400) > $OUT/gputest.log; cat $OUT/gputest.log
for rep in 1 2 3; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-270; done; done > $OUT/ab_early_occ.txt
cat $OUT/ab_early_occ.txt
