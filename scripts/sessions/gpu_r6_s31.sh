#!/bin/bash
# Round 6 session 31: the tree (A) now launches the stroke kernel with 384 and k_path_count's in-flight form with 768 workgroups (session 30: +1.1 %).  More grids
# with frames in flight: k_path_tiling 1024 (K) / 4096 (L; tree 2048), k_backdrop's cap 2048 (M) / 1024 (N; tree 8192), the heavy flatten workgroups 1024 (R) / 512 (S; tree 2048)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s31
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2 3; do for L in A K L M N R S; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-60; done; done > $OUT/ab_in_flight_grids2.txt
cat $OUT/ab_in_flight_grids2.txt
