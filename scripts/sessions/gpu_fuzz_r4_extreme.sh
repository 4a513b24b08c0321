#!/bin/bash
# The extreme-value part of the round 4 GPU campaign on its own (the first attempt gave the three runs 170 s, and the oracle
# alone takes ~1 s a seed here: a few seeds in a hundred emit 10-20 M lines and grow its pools to 16x): product library on the
# GPU against the oracle with growable pools, default flags / stroke kernel forced / fine's slices forced.
#     T=420 bash scripts/gpu_fuzz_r4_extreme.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-420}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run extreme 3000 3090
FUZZ_STROKE_KERNEL=1 run extreme 3250 3340
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=2 run extreme 3500 3590
wait
