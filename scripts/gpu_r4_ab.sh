#!/bin/bash
# round 4: same-box A/B of the in-tree build (A) against ab_tmp/libvello_hip_<X>.so for X in $VARIANTS, plus the SQ instruction
# counters of k_fine for each (one rocprofv3 --pmc pass per library, kernel trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4ab
REPS="${REPS:-1 2}" STEPS=${STEPS:-80} bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r4ab/ab.txt
if [ -n "${PMC:-}" ]; then
for w in A ${VARIANTS:-B}; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/r4ab/pmc_$w -o p -- python scripts/ab_bench.py $w --steps 10 --warmup 2 --in-flight 1 --timed-only > gpurun_out/r4ab/pmc_$w.log 2>&1
  f=$(find gpurun_out/r4ab/pmc_$w -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > gpurun_out/r4ab/pmc_$w.summary.txt; grep -A8 "k_fine" gpurun_out/r4ab/pmc_$w.summary.txt | head -12; rm -rf gpurun_out/r4ab/pmc_$w; else tail -5 gpurun_out/r4ab/pmc_$w.log; fi
done
fi
