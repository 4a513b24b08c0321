#!/bin/bash
# A differential fuzz campaign on the GPU (product library against the oracle), several processes side by side.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-170}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run api 20000 30000
run api 30000 40000
FUZZ_STROKE_KERNEL=1 run api 40000 50000
FUZZ_STROKE_KERNEL=1 run api 50000 60000
run sizes 3000 6000
FUZZ_STROKE_KERNEL=1 run sizes 6000 9000
run pools 2000 3500
FUZZ_STROKE_KERNEL=1 run pools 3500 5000
run extreme 400 700
FUZZ_STROKE_KERNEL=1 run extreme 700 1000
wait
