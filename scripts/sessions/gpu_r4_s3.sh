#!/bin/bash
# round 4, session 3: v_cvt saturation check, the GPU suite at the new conversions, A/B against the round's start
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s3
scripts/calib/cvt_sat 2>&1 | tee gpurun_out/r4s3/cvt_sat.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r4s3/gputest.txt
VARIANTS="BASE T3" PMC=1 bash scripts/gpu_r4_ab.sh
