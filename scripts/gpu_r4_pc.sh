#!/bin/bash
# SQ counters of k_path_count for the in-tree build (A) and ab_tmp/libvello_hip_<X>.so (VARIANTS), one frame at a time, d2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4pc
for w in A ${VARIANTS:-R4}; do
  for pass in 1 2; do
    if [ $pass = 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; else C="SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; fi
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/r4pc/pmc_$w$pass -o p -- python scripts/ab_bench.py $w --steps 10 --warmup 2 --in-flight 1 --timed-only > gpurun_out/r4pc/pmc_$w$pass.log 2>&1
    f=$(find gpurun_out/r4pc/pmc_$w$pass -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > gpurun_out/r4pc/pmc_$w$pass.summary.txt; echo "== $w pass $pass"; python scripts/pmc_clusters.py "$f" k_path_count; rm -rf gpurun_out/r4pc/pmc_$w$pass; else tail -5 gpurun_out/r4pc/pmc_$w$pass.log; fi
  done
done
