#!/bin/bash
# Round 4, third GPU campaign at the round's last kernel commit (k_path_count: 768 striding workgroups): longer runs, fresh seeds.
#     T=270 bash scripts/gpu_fuzz_r4c.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-270}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run api 400000 416000
FUZZ_IN_FLIGHT=4 run api 416000 432000
FUZZ_STROKE_KERNEL=1 run api 432000 448000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 FUZZ_FINE_SLICES=1 run api 448000 464000
run sizes 50000 55000
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=3 run sizes 55000 60000
run pools 30000 33000
FUZZ_STROKE_KERNEL=1 run pools 33000 36000
run extreme 5000 5100
FUZZ_FINE_SLICES=1 run extreme 5100 5200
wait
