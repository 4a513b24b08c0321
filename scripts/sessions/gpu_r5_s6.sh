#!/bin/bash
# round 5, session 6: A = tree (waves choose between walking together and each lane on its own; ms_fill_simple's idle lanes on
# pixels of their own) against R4: flatten's kernels, bench A/B with k_fine's SQ instruction counters, the MSAA / flatten GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tiger or mmark or stroke or cardioid or d2 or c3 or random or slices or msaa or fuzz or smoke" 2>&1 | tail -3 | tee $O/tests.txt
for v in A R4 A R4; do timeout 300 python scripts/flatten_kernels.py $v 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
VARIANTS="R4" PMC=1 REPS="1 2" STEPS=100 bash scripts/gpu_r4_ab.sh 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
cp gpurun_out/r4ab/pmc_*.summary.txt $O/
