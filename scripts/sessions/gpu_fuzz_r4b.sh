#!/bin/bash
# Round 4, second GPU campaign: after k_path_count became the LDS-table kernel (and k_backdrop's block size a shift).  Fresh seed
# ranges; the extreme-value mode against the oracle with growable pools.    T=150 bash scripts/gpu_fuzz_r4b.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-150}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run api 300000 306000
FUZZ_IN_FLIGHT=3 run api 306000 312000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run api 312000 318000
run sizes 40000 42000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run sizes 42000 44000
run pools 20000 21000
FUZZ_IN_FLIGHT=2 run pools 21000 22000
run extreme 4000 4060
FUZZ_STROKE_KERNEL=1 run extreme 4060 4120
wait
