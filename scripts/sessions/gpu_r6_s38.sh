#!/bin/bash
# Round 6 session 38: k_backdrop with the rows' sums by DPP (row shifts + two broadcasts) instead of six ds_bpermute rounds -- tree (A) against HEAD (H)
# k_flatten_main one frame at a time, where a workgroup walks ONE round; with frames in flight k_flatten_strokes' 384 workgroups walk seven).  Tree (A) against HEAD (H).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s38
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "config_c or back_half or tiger or catalogue or fuzz or backdrop" 2>&1 | tail -2) > $OUT/gputest.log; cat $OUT/gputest.log
for rep in 1 2 3 4; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-270; done; done > $OUT/ab_backdrop_dpp.txt
cat $OUT/ab_backdrop_dpp.txt
