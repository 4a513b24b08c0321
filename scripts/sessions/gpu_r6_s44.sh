#!/bin/bash
# Round 6 session 44: early occlusion with the fills' phase on a list of their lines, the table's words a step ahead, k_occ_build's loads together -- tree (A) against HEAD (H);
# kernels one frame at a time by rocprofv3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s44
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "early_occ or config_c or back_half or tiger or catalogue or fuzz or path_count" 2>&1 | tail -30 | cut -c1-400) > $OUT/gputest.log; tail -3 $OUT/gputest.log
for rep in 1 2 3; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-270; done; done > $OUT/ab_early_occ.txt
cat $OUT/ab_early_occ.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp -o p -- python bench.py --workload d2 --steps 50 --warmup 5 --in-flight 1 --timed-only > $OUT/serial.log 2>&1
find $OUT/tmp -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats_serial_d2.csv
rm -rf $OUT/tmp
