#!/bin/bash
# round 4, session 5: the GPU suite with coarse's inputs made by k_backdrop, A/B against the build before (T5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4s5
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r4s5/gputest.txt
VARIANTS="T5" bash scripts/gpu_r4_ab.sh
