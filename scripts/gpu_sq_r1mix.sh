#!/bin/bash
# the rest of the per-commit PMC evidence after scripts/gpu_head_check.sh: SQ counters on d2, traffic passes on r1mix
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02}
OUT=gpurun_out/head2
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
pmc() { # name counters -- cmd...
  name=$1; shift; ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout 40 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $OUT/tmp_$name -o p -- "$@" > $OUT/$name.log 2>&1
  f=$(find $OUT/tmp_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > $OUT/${TAG}_pmc_$name.summary.txt; else tail -3 $OUT/$name.log; fi
  rm -rf $OUT/tmp_$name
}
CMD="python bench.py --workload d2 --steps 8 --warmup 2 --in-flight 1 --timed-only"
pmc sq1_d2 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -- $CMD
pmc sq2_d2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- $CMD
CMD="python bench.py --workload r1mix --steps 8 --warmup 2 --in-flight 1 --timed-only"
pmc fetch_r1mix FETCH_SIZE -- $CMD
pmc write_r1mix WRITE_SIZE -- $CMD
rm -f $OUT/*.log
ls $OUT
