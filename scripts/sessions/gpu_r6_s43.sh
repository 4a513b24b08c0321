#!/bin/bash
# Round 6 session 43: early occlusion -- the kernels of the two-phase stage one frame at a time (rocprofv3 --stats), d2
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s43
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -o p -- python bench.py --workload d2 --steps 50 --warmup 5 --in-flight 1 --timed-only > $OUT/serial.log 2>&1
find $OUT/tmp -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats_serial_d2.csv
rm -rf $OUT/tmp
cut -d, -f1-4 $OUT/kernel_stats_serial_d2.csv | head -30
