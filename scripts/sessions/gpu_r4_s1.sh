#!/bin/bash
# round 4, session 1: instruction issue costs (scripts/calib/valu_rate.hip) + the round's starting point on this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s1
timeout 120 scripts/calib/valu_rate > gpurun_out/r4s1/valu_rate.txt 2>&1
cat gpurun_out/r4s1/valu_rate.txt
STEPS=60 bash scripts/gpu_quick.sh > gpurun_out/r4s1/quick.txt 2>&1
cat gpurun_out/r4s1/quick.txt
