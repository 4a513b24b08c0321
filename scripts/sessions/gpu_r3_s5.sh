#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s5
mkdir -p $OUT
timeout 120 python scripts/fine_timeline.py d2 > $OUT/timeline_slices_d2.txt 2>&1; head -70 $OUT/timeline_slices_d2.txt
VELLO_PROF_LIB=ab_tmp/libvello_hip_TLZ.so timeout 120 python scripts/fine_timeline.py d2 r1mix > $OUT/timeline_noslices.txt 2>&1; head -70 $OUT/timeline_noslices.txt
