#!/bin/bash
# Round 5, second sitting: differential fuzz on the GPU at the round's last kernel commit (coarse's tile bits a word per 8 tiles; k_front;
# k_path_count's long lines): fresh seed ranges, the engine's own choices and the forced paths.    T=90 bash scripts/sessions/gpu_fuzz_r5b.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export FUZZ_GPU=1
T=${T:-90}
O=gpurun_out/r5b_fuzz
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -3 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
{
run api 730000 740000
FUZZ_IN_FLIGHT=3 run api 740000 750000
FUZZ_FLATTEN=coop FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 FUZZ_FINE_SLICES=1 run api 750000 760000
run sizes 53000 56000
FUZZ_STROKE_KERNEL=1 FUZZ_FINE_SLICES=1 run sizes 56000 59000
run pools 61000 62500
FUZZ_FLATTEN=alone run extreme 6200 6400
wait
} | tee $O/r05_gpu_fuzz_second_sitting.txt
