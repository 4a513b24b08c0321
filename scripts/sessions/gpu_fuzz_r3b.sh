#!/bin/bash
# Round 3, second differential fuzz campaign on the GPU: both launch shapes of flatten's stroke workgroups -- beside the heavy
# list's (one frame in flight, the default) and as a kernel of their own (FUZZ_IN_FLIGHT=2) -- forced on for every scene
# (VELLO_HIP_DEBUG_STROKE_KERNEL), fresh seed ranges.    T=170 bash scripts/gpu_fuzz_r3b.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-170}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
FUZZ_STROKE_KERNEL=1 run api 100000 110000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run api 110000 120000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run api 120000 130000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=3 FUZZ_FINE_SLICES=1 run api 130000 140000
FUZZ_STROKE_KERNEL=1 run sizes 15000 18000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run sizes 18000 21000
FUZZ_STROKE_KERNEL=1 run pools 8000 9500
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 9500 11000
FUZZ_STROKE_KERNEL=1 run extreme 1600 1900
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run extreme 1900 2200
wait
