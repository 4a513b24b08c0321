#!/bin/bash
# Round 6: the evidence for the round's LAST kernel commit in one GPU session: the GPU suite, the bench line as the driver runs it
# and with the defaults, rocprofv3 kernel stats (one frame at a time d2 / r1mix / mmark / tiger, four in flight d2), PMC traffic with
# its calibration and the SQ counters on d2, the kernel time line with four frames in flight, the other workloads, the kernels'
# resources from the code objects.
# Every step under its own timeout, results in gpurun_out/r6_evidence/ as they come.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=r06
OUT=gpurun_out/r6_evidence
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/${TAG}_gputest.log; tail -2 $OUT/${TAG}_gputest.log
timeout 200 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_flags_k20.json; head -c 260 $OUT/${TAG}_bench_driver_flags_k20.json; echo
stats() { # name cmd...
  name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -o p -- "$@" > $OUT/$name.log 2>&1
  find $OUT/tmp_$name -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $OUT/${TAG}_kernel_stats_$name.csv
  rm -rf $OUT/tmp_$name
}
pmc() { # name counters -- cmd...
  name=$1; shift; ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/tmp_$name -o p -- "$@" > $OUT/$name.log 2>&1
  f=$(find $OUT/tmp_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > $OUT/${TAG}_pmc_$name.summary.txt; else tail -5 $OUT/$name.log; fi
  rm -rf $OUT/tmp_$name
}
stats serial_d2 python bench.py --workload d2 --steps 50 --warmup 5 --in-flight 1 --timed-only
stats serial_r1mix python bench.py --workload r1mix --steps 50 --warmup 5 --in-flight 1 --timed-only
stats serial_mmark python scripts/render_loop.py mmark 40 1
stats serial_tiger python scripts/render_loop.py tiger 40 1
stats pipelined_d2 python bench.py --workload d2 --steps 100 --warmup 10 --timed-only
if [ ! -x scripts/calib/pmc_calib ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/calib/pmc_calib.hip -o scripts/calib/pmc_calib; fi
pmc calib_fetch FETCH_SIZE -- scripts/calib/pmc_calib
pmc calib_write WRITE_SIZE -- scripts/calib/pmc_calib
CMD="python bench.py --workload d2 --steps 8 --warmup 2 --in-flight 1 --timed-only"
pmc fetch_d2 FETCH_SIZE -- $CMD
pmc write_d2 WRITE_SIZE -- $CMD
CMDR="python bench.py --workload r1mix --steps 8 --warmup 2 --in-flight 1 --timed-only"
pmc fetch_r1mix FETCH_SIZE -- $CMDR
pmc write_r1mix WRITE_SIZE -- $CMDR
pmc sq1_d2 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -- $CMD
pmc sq2_d2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- $CMD
python scripts/make_pmc_traffic.py $OUT $TAG > $OUT/pmc_traffic.json 2> $OUT/make_pmc_traffic.err; head -c 300 $OUT/pmc_traffic.json; echo
# (the defaults' bench line AFTER the PMC passes, with this session's traffic file in place: traffic_stale false)
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout 300 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json; head -c 260 $OUT/${TAG}_bench.json; echo
NIF="4 1" timeout 300 bash scripts/gpu_r4_timeline.sh > $OUT/${TAG}_pipeline_timeline.txt 2>&1; grep -A3 "window" $OUT/${TAG}_pipeline_timeline.txt | head -8
python scripts/kernel_resources.py > $OUT/${TAG}_kernel_resources.txt 2>&1; head -3 $OUT/${TAG}_kernel_resources.txt
timeout 200 python scripts/other_workloads.py 2>/dev/null > $OUT/${TAG}_other_workloads.jsonl; wc -l $OUT/${TAG}_other_workloads.jsonl
rm -f $OUT/*.log.tmp
ls $OUT
