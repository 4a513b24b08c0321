#!/bin/bash
# Round 4, the last GPU seconds: api fuzz in four flag settings at the last kernel commit (17 s each).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
run() { ( timeout 17 python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run api 510000 511800
FUZZ_IN_FLIGHT=4 FUZZ_STROKE_KERNEL=1 run api 511800 513600
FUZZ_IN_FLIGHT=2 FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run api 513600 515400
FUZZ_STROKE_KERNEL=1 run api 515400 517200
wait
