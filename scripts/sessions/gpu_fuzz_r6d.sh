#!/bin/bash
# Round 6, fourth campaign: differential fuzz on the GPU at the round's LAST kernel commit (100685c: the stroke kernel hands untame inputs to the heavy code),
# the stroke kernel forced in most processes, new seed ranges, six processes at a time, two waves.    T=300 bash scripts/sessions/gpu_fuzz_r6d.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export FUZZ_GPU=1
T=${T:-300}
O=gpurun_out/r6_fuzz_d
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -2 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
{
FUZZ_STROKE_KERNEL=1 run api 6000000 6100000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run api 6100000 6200000
FUZZ_STROKE_KERNEL=1 FUZZ_FLATTEN=alone FUZZ_IN_FLIGHT=2 run api 6200000 6300000
FUZZ_STROKE_KERNEL=1 FUZZ_FLATTEN=coop run sizes 800000 830000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 830000 845000
FUZZ_STROKE_KERNEL=1 run extreme 30000 32000
wait
FUZZ_IN_FLIGHT=4 run api 6300000 6400000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run api 6400000 6500000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run sizes 845000 875000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run extreme 32000 34000
run pools 875000 890000
FUZZ_FLATTEN=coop FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=3 run api 6500000 6600000
wait
} | tee $O/r06_gpu_fuzz_d.txt
