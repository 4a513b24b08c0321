#!/bin/bash
# Round 6 session 36: the stroke workgroups with points / style / transform one round ahead as well (tags two, list entries three) -- tree (A) against HEAD (H: tags one, entries two)
# k_flatten_main one frame at a time, where a workgroup walks ONE round; with frames in flight k_flatten_strokes' 384 workgroups walk seven).  Tree (A) against HEAD (H).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s36
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "stroke or config_c3 or flatten or c5" 2>&1 | tail -2) > $OUT/gputest.log; cat $OUT/gputest.log
for rep in 1 2 3 4; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-140; done; done > $OUT/ab_stroke_prefetch3.txt
cat $OUT/ab_stroke_prefetch3.txt
IN_FLIGHT=4 python scripts/stroke_timeline.py d2 2>&1 | grep -v amdgpu.ids | tee $OUT/stroke_timeline_d2.txt
