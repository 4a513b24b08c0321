#!/bin/bash
# Round 3's differential fuzz campaign on the GPU: fresh seed ranges, every tile through fine's sliced path
# (VELLO_HIP_DEBUG_FINE_SLICES) and / or the stroke kernel forced (arcs set aside for the heavy kernel), several processes
# side by side.    T=170 bash scripts/gpu_fuzz_r3.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-170}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
FUZZ_FINE_SLICES=1 run api 60000 70000
FUZZ_FINE_SLICES=1 FUZZ_STROKE_KERNEL=1 run api 70000 80000
FUZZ_STROKE_KERNEL=1 run api 80000 90000
run api 90000 100000
FUZZ_FINE_SLICES=1 run sizes 9000 12000
FUZZ_FINE_SLICES=1 FUZZ_STROKE_KERNEL=1 run sizes 12000 15000
FUZZ_FINE_SLICES=1 run pools 5000 6500
FUZZ_STROKE_KERNEL=1 run pools 6500 8000
FUZZ_FINE_SLICES=1 FUZZ_STROKE_KERNEL=1 run extreme 1000 1300
run extreme 1300 1600
wait
