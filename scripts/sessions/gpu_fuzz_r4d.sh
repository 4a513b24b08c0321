#!/bin/bash
# Round 4, last GPU seconds: a short campaign at the round's last kernel commit (the merged reservations of k_flatten_light, k_coarse and
# the stroke workgroups, the striding stroke grid).    T=35 bash scripts/gpu_fuzz_r4d.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-35}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run api 500000 501500
FUZZ_IN_FLIGHT=3 FUZZ_STROKE_KERNEL=1 run api 501500 503000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run api 503000 504500
FUZZ_IN_FLIGHT=2 FUZZ_STROKE_KERNEL=1 run sizes 60000 60400
FUZZ_STROKE_KERNEL=1 run pools 40000 40300
wait
