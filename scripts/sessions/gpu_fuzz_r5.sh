#!/bin/bash
# Round 5 differential fuzz campaign on the GPU at the round's last kernel commit (the wave-cooperative and level-parallel flatten,
# both sets of heavy-list kernels, ms_fill_simple, rare_command inlined into the brush kernels): the engine's own choice of flatten
# kernels, each set forced, the stroke workgroups forced in both launch shapes, fine's slices forced; the extreme-value mode against
# the oracle with growable pools.  Fresh seed ranges.    T=120 bash scripts/sessions/gpu_fuzz_r5.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export FUZZ_GPU=1
T=${T:-120}
# (a run cut off by the timeout reports how far it came: fuzz_campaign.py prints "at <seed>" every 500 seeds)
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -4 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
run api 600000 610000
FUZZ_FLATTEN=coop run api 610000 620000
FUZZ_FLATTEN=alone FUZZ_IN_FLIGHT=3 run api 620000 630000
FUZZ_FLATTEN=coop FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 FUZZ_FINE_SLICES=1 run api 630000 640000
FUZZ_FLATTEN=coop run sizes 40000 43000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run sizes 43000 46000
FUZZ_FLATTEN=coop run pools 20000 21500
FUZZ_FLATTEN=alone FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 21500 23000
FUZZ_FLATTEN=coop run extreme 4000 4250
FUZZ_FLATTEN=alone run extreme 4250 4500
FUZZ_FINE_SLICES=1 run extreme 4500 4700
wait
