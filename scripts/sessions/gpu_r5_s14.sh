#!/bin/bash
# round 5, session 14: a wave with one curve walks it level-parallel (63 ranges of the dyadic tree a turn): flatten's kernels against
# R4 and against A0 (the last commit: lockstep only), the flatten GPU tests, the other workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "flatten_kernel_sets or tiger or stroke or cardioid or tricky or funky or robust or fuzz or smoke or circle" 2>&1 | tail -3 | tee $O/tests.txt
for v in A A0 R4 A A0 R4; do timeout 300 python scripts/flatten_kernels.py $v tiger mmark d2 2>/dev/null | grep -v amdgpu.ids | tee -a $O/flatten_kernels.txt; done
timeout 300 python scripts/flatten_prof.py tiger 2>&1 | grep -v amdgpu.ids | tee $O/flatten_prof.txt
timeout 300 python scripts/other_workloads.py 2>/dev/null | tee $O/other_workloads.jsonl
