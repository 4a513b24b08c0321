#!/bin/bash
# Round 5, second sitting: a second differential fuzz pass on the GPU at the round's last kernel commit, other flag settings and seed ranges than
# gpu_fuzz_r5b.sh.    T=70 bash scripts/sessions/gpu_fuzz_r5c.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export FUZZ_GPU=1
T=${T:-70}
O=gpurun_out/r5c_fuzz
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -2 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
{
FUZZ_FLATTEN=alone run api 790000 800000
FUZZ_FLATTEN=coop FUZZ_IN_FLIGHT=4 run api 800000 810000
FUZZ_STROKE_KERNEL=1 run api 810000 820000
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=2 run api 820000 830000
FUZZ_FLATTEN=coop run sizes 62000 65000
FUZZ_FLATTEN=alone FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 65000 66500
wait
} | tee $O/r05_gpu_fuzz_second_sitting_b.txt
