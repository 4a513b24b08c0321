#!/bin/bash
# Round 3, third differential fuzz campaign on the GPU, at the round's last kernel commit (PATH-marker fields written by the
# pathtag scan, the draw stage's workgroups in k_flatten_light's launch): default flags, the stroke workgroups forced in both
# launch shapes, fine's slices forced; fresh seed ranges.    T=170 bash scripts/gpu_fuzz_r3c.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-170}
run() { ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done" | tail -6 ) & }
run api 140000 150000
FUZZ_IN_FLIGHT=3 run api 150000 160000
FUZZ_STROKE_KERNEL=1 run api 160000 170000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 FUZZ_FINE_SLICES=1 run api 170000 180000
run sizes 21000 24000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run sizes 24000 27000
run pools 11000 12500
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 12500 14000
run extreme 2200 2400
FUZZ_STROKE_KERNEL=1 run extreme 2400 2600
wait
