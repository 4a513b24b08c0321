#!/bin/bash
# Round 4 differential fuzz campaign on the GPU at the round's last kernel commit (fine's per-batch row table, SWAR coverage,
# per-segment edge masks, the bare v_cvt conversions in every kernel, backdrop's zero-step skip): default flags, the stroke
# workgroups forced in both launch shapes, fine's slices forced; the extreme-value mode against the oracle with growable pools
# (VERDICT r3 item 9).  Fresh seed ranges.    T=170 bash scripts/gpu_fuzz_r4.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-170}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run api 200000 210000
FUZZ_IN_FLIGHT=3 run api 210000 220000
FUZZ_STROKE_KERNEL=1 run api 220000 230000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 FUZZ_FINE_SLICES=1 run api 230000 240000
run sizes 30000 33000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run sizes 33000 36000
run pools 15000 16500
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 16500 18000
run extreme 3000 3250
FUZZ_STROKE_KERNEL=1 run extreme 3250 3500
FUZZ_FINE_SLICES=1 run extreme 3500 3700
wait
