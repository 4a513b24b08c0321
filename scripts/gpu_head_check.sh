#!/bin/bash
# One short GPU session for a commit: the whole GPU suite, the bench line, rocprofv3 kernel stats (one frame at a time, d2) and the
# d2 PMC traffic passes with their calibration -- in that order, each step under its own timeout, so that a session that is cut
# short still leaves the earlier results in gpurun_out/head/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02}
OUT=gpurun_out/head
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 150 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $OUT/gputest.log; tail -2 $OUT/gputest.log
timeout 90 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_head.json; head -c 300 $OUT/${TAG}_bench_head.json; echo
pmc() { # name counter -- cmd...
  name=$1; ctr=$2; shift 3
  timeout 60 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/tmp_$name -o p -- "$@" > $OUT/$name.log 2>&1
  f=$(find $OUT/tmp_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > $OUT/${TAG}_pmc_$name.summary.txt; else tail -3 $OUT/$name.log; fi
  rm -rf $OUT/tmp_$name
}
CMD="python bench.py --workload d2 --steps 8 --warmup 2 --in-flight 1 --timed-only"
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_stats -o p -- python bench.py --workload d2 --steps 30 --warmup 5 --in-flight 1 --timed-only > $OUT/stats.log 2>&1
find $OUT/tmp_stats -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/${TAG}_kernel_stats_serial_d2_head.csv; rm -rf $OUT/tmp_stats
pmc fetch_d2 FETCH_SIZE -- $CMD
pmc write_d2 WRITE_SIZE -- $CMD
if [ ! -x scripts/calib/pmc_calib ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/calib/pmc_calib.hip -o scripts/calib/pmc_calib; fi
pmc calib_fetch FETCH_SIZE -- scripts/calib/pmc_calib
pmc calib_write WRITE_SIZE -- scripts/calib/pmc_calib
python scripts/make_pmc_traffic.py $OUT $TAG > $OUT/pmc_traffic_d2.json
rm -f $OUT/*.log.tmp
ls $OUT
