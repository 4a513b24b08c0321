#!/bin/bash
# Everything profiles/ holds for one state of the code, in one GPU session: rocprofv3 kernel stats (one frame at a time,
# both workloads; the default frames in flight for the agreement check), PMC traffic (calibrated FETCH_SIZE / WRITE_SIZE, separate
# passes, kernel trace only) and SQ counters.  Results land in gpurun_out/profiles_<tag>/; copy them to profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
stats() { # name cmd...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -o p -- "$@" > $OUT/$name.log 2>&1
  find $OUT/tmp_$name -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/${TAG}_kernel_stats_$name.csv
  rm -rf $OUT/tmp_$name
}
pmc() { # name counters -- cmd...
  name=$1; shift; ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/tmp_$name -o p -- "$@" > $OUT/$name.log 2>&1
  f=$(find $OUT/tmp_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > $OUT/${TAG}_pmc_$name.summary.txt; else tail -5 $OUT/$name.log; fi
  rm -rf $OUT/tmp_$name
}
for wl in d2 r1mix; do
  stats serial_$wl python bench.py --workload $wl --steps 50 --warmup 5 --in-flight 1 --timed-only
done
stats pipelined_d2 python bench.py --workload d2 --steps 100 --warmup 10 --timed-only
python bench.py --workload d2 --steps 100 --warmup 10 --timed-only 2>/dev/null | tail -1 > $OUT/${TAG}_bench_timed_only_d2.json
if [ ! -x scripts/calib/pmc_calib ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/calib/pmc_calib.hip -o scripts/calib/pmc_calib; fi
pmc calib_fetch FETCH_SIZE -- scripts/calib/pmc_calib
pmc calib_write WRITE_SIZE -- scripts/calib/pmc_calib
for wl in d2 r1mix; do
  CMD="python bench.py --workload $wl --steps 8 --warmup 2 --in-flight 1 --timed-only"
  pmc fetch_$wl FETCH_SIZE -- $CMD
  pmc write_$wl WRITE_SIZE -- $CMD
  pmc sq1_$wl SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -- $CMD
  pmc sq2_$wl SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- $CMD
done
git rev-parse --short HEAD > $OUT/commit.txt 2>/dev/null || cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
python scripts/make_pmc_traffic.py $OUT $TAG > $OUT/pmc_traffic.json
rm -f $OUT/*.log
ls $OUT
