#!/bin/bash
# Round 6 session 12: vello_hip_create makes the null stream take the first hardware queue: the engine created first (ring_last, bench.py's order) should now reach the 2 300 level
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s12
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(for m in ring_last tiny_first ring_last; do timeout 200 python scripts/bench_loop_probe.py $m 2>/dev/null; done
 echo "== R5 library (ab_tmp/libvello_hip_R5.so), ring_last"; PROBE_LIBRARY=ab_tmp/libvello_hip_R5.so timeout 200 python scripts/bench_loop_probe.py ring_last 2>/dev/null | grep '"bare"'
) > $OUT/first_queue_fix.txt
cat $OUT/first_queue_fix.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_k20.json; head -c 300 $OUT/bench_k20.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_k20_b.json; head -c 300 $OUT/bench_k20_b.json; echo
