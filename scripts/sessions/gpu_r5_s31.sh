#!/bin/bash
# round 5, session 31: settings of the HIP / HSA runtime under the same library -- kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1),
# completion signals polled instead of interrupt-driven (HSA_ENABLE_INTERRUPT=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s31
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
{
env FOO=1 timeout 100 python scripts/sessions/gpu_r5_s31.py default
env HIP_FORCE_DEV_KERNARG=1 timeout 100 python scripts/sessions/gpu_r5_s31.py dev_kernarg
env HSA_ENABLE_INTERRUPT=0 timeout 100 python scripts/sessions/gpu_r5_s31.py no_interrupt
env HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0 timeout 100 python scripts/sessions/gpu_r5_s31.py both
env FOO=1 timeout 100 python scripts/sessions/gpu_r5_s31.py default
} 2>/dev/null | tee $O/out.txt
