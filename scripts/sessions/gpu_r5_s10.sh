#!/bin/bash
# round 5, session 10: why is k_flatten_main slower than R4's on r1mix / d2?  SQ + instruction-cache counters of both libraries
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s10
mkdir -p $O
for w in A R4; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU"; do
    n=$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${w}_$n -o p -- python scripts/flatten_kernels.py $w r1mix d2 > $O/pmc_${w}_$n.log 2>&1
    f=$(find $O/pmc_${w}_$n -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > $O/pmc_${w}_$n.summary.txt; rm -rf $O/pmc_${w}_$n; else tail -3 $O/pmc_${w}_$n.log; fi
  done
done
grep -h -A10 "k_flatten_main" $O/pmc_*.summary.txt | head -120
