#!/bin/bash
# round 4, session 2: instruction issue costs, 256-thread workgroups (scripts/calib/valu_rate.hip)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s2
timeout 200 scripts/calib/valu_rate > gpurun_out/r4s2/valu_rate.txt 2>&1
cat gpurun_out/r4s2/valu_rate.txt
