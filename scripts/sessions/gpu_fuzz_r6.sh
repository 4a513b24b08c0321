#!/bin/bash
# Round 6: differential fuzz on the GPU at the round's last kernel commit (k_path_count's flush in groups of four turns is what changed since the campaign
# of round 5): new seed ranges, six processes at a time, two waves of them.    T=150 bash scripts/sessions/gpu_fuzz_r6.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export FUZZ_GPU=1
T=${T:-150}
O=gpurun_out/r6_fuzz
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -2 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
{
FUZZ_FLATTEN=alone run api 900000 920000
FUZZ_FLATTEN=coop FUZZ_IN_FLIGHT=4 run api 920000 940000
FUZZ_STROKE_KERNEL=1 run api 940000 960000
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=2 run api 960000 980000
FUZZ_FLATTEN=coop run sizes 70000 76000
FUZZ_FLATTEN=alone FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 76000 79000
wait
FUZZ_IN_FLIGHT=4 run api 980000 1000000
FUZZ_FLATTEN=alone FUZZ_IN_FLIGHT=3 FUZZ_STROKE_KERNEL=1 run api 1000000 1020000
FUZZ_FLATTEN=coop FUZZ_FINE_SLICES=1 run sizes 80000 86000
run extreme 7000 7400
FUZZ_IN_FLIGHT=2 run pools 86000 89000
FUZZ_FLATTEN=alone run sizes 90000 96000
wait
} | tee $O/r06_gpu_fuzz.txt
