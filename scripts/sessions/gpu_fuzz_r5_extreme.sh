#!/bin/bash
# Round 5, the extreme-value mode on the GPU (coordinates up to f32's limits, +-inf, NaN; the oracle with growable pools) in four flag
# settings: the engine's choice of flatten kernels, each set forced (the cooperative walk sends untame curves to the plain one), slices.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-240}
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -3 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
run extreme 4000 4250
FUZZ_FLATTEN=coop run extreme 4250 4500
FUZZ_FLATTEN=alone FUZZ_STROKE_KERNEL=1 run extreme 4500 4750
FUZZ_FLATTEN=coop FUZZ_FINE_SLICES=1 run extreme 4750 5000
wait
