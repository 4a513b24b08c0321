#!/bin/bash
# Round 6 session 40: frames in flight 3 / 4 / 5 / 6 / 8 with the round's kernels (the last sweep, r05_inflight_sweep.txt, predates the
# small in-flight forms of k_path_count / k_fine and the queue-order fix), a process each, with 8 and 16 hardware queues
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s40
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
for rep in 1 2; do for q in 8 16; do for nif in 4 3 5 6 8; do
  echo -n "hwq $q nif $nif  "; GPU_MAX_HW_QUEUES=$q VELLO_AB_NIF=$nif timeout 120 python scripts/ab_process.py A d2 mmark 2>/dev/null | cut -c1-60 | tr '\n' ' '; echo
done; done; done > $OUT/inflight_sweep.txt
cat $OUT/inflight_sweep.txt
