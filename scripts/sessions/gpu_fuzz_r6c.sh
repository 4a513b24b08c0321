#!/bin/bash
# Round 6, third campaign (longer processes, new seed ranges): differential fuzz on the GPU at the round's LAST kernel commit (since the first campaign at 92e1f73: k_backdrop's DPP row sums and
# clamped requests, the stroke workgroups' load order and box atomics, wave_bbox_update's early out, the in-flight grids): new seed ranges, six processes at a
# time, three waves.    T=240 bash scripts/sessions/gpu_fuzz_r6b.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export FUZZ_GPU=1
T=${T:-560}
O=gpurun_out/r6_fuzz_c
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -2 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
{
FUZZ_FLATTEN=alone run api 3100000 3190000
FUZZ_FLATTEN=coop FUZZ_IN_FLIGHT=4 run api 3130000 3220000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run api 3160000 3250000
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=2 run api 3190000 3280000
FUZZ_FLATTEN=coop run sizes 400000 430000
FUZZ_FLATTEN=alone FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 410000 425000
wait
FUZZ_IN_FLIGHT=4 run api 3220000 3310000
FUZZ_FLATTEN=alone FUZZ_IN_FLIGHT=3 FUZZ_STROKE_KERNEL=1 run api 3250000 3340000
FUZZ_FLATTEN=coop FUZZ_FINE_SLICES=1 run sizes 415000 445000
run extreme 11000 12800
FUZZ_IN_FLIGHT=4 run pools 425000 440000
FUZZ_FLATTEN=alone FUZZ_IN_FLIGHT=4 run sizes 430000 460000
wait
FUZZ_STROKE_KERNEL=1 run api 3280000 3370000
FUZZ_FLATTEN=coop FUZZ_IN_FLIGHT=2 run api 3310000 3400000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run sizes 440000 470000
FUZZ_STROKE_KERNEL=1 run extreme 11600 13400
FUZZ_FLATTEN=coop run pools 450000 465000
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=4 run api 3340000 3430000
wait
} | tee $O/r06_gpu_fuzz_c.txt
