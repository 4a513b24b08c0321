#!/bin/bash
# PMC pass over one workload, one frame at a time (separate from --stats runs; counters only with --kernel-trace).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
WL=${WL:-d2}
CMD="python bench.py --workload $WL --steps 6 --warmup 2 --in-flight 1 --timed-only"
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -o p -- $CMD > gpurun_out/pmc/$name.log 2>&1
  f=$(find gpurun_out/pmc/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > gpurun_out/pmc/${WL}_$name.summary.txt; grep -A12 "${KERNEL:-k_coarse}" gpurun_out/pmc/${WL}_$name.summary.txt | head -${LINES_OUT:-26}; rm -rf gpurun_out/pmc/$name; else tail -5 gpurun_out/pmc/$name.log; fi
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
