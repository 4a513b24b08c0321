#!/bin/bash
# Round 6, second campaign: differential fuzz on the GPU at the round's LAST kernel commit (since the first campaign at 92e1f73: k_backdrop's DPP row sums and
# clamped requests, the stroke workgroups' load order and box atomics, wave_bbox_update's early out, the in-flight grids): new seed ranges, six processes at a
# time, three waves.    T=240 bash scripts/sessions/gpu_fuzz_r6b.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export FUZZ_GPU=1
T=${T:-240}
O=gpurun_out/r6_fuzz_b
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -2 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} sl=${FUZZ_FINE_SLICES:-0} $*] /" ) & }
{
FUZZ_FLATTEN=alone run api 1100000 1130000
FUZZ_FLATTEN=coop FUZZ_IN_FLIGHT=4 run api 1130000 1160000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run api 1160000 1190000
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=2 run api 1190000 1220000
FUZZ_FLATTEN=coop run sizes 100000 110000
FUZZ_FLATTEN=alone FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run pools 110000 115000
wait
FUZZ_IN_FLIGHT=4 run api 1220000 1250000
FUZZ_FLATTEN=alone FUZZ_IN_FLIGHT=3 FUZZ_STROKE_KERNEL=1 run api 1250000 1280000
FUZZ_FLATTEN=coop FUZZ_FINE_SLICES=1 run sizes 115000 125000
run extreme 8000 8600
FUZZ_IN_FLIGHT=4 run pools 125000 130000
FUZZ_FLATTEN=alone FUZZ_IN_FLIGHT=4 run sizes 130000 140000
wait
FUZZ_STROKE_KERNEL=1 run api 1280000 1310000
FUZZ_FLATTEN=coop FUZZ_IN_FLIGHT=2 run api 1310000 1340000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run sizes 140000 150000
FUZZ_STROKE_KERNEL=1 run extreme 8600 9200
FUZZ_FLATTEN=coop run pools 150000 155000
FUZZ_FINE_SLICES=1 FUZZ_IN_FLIGHT=4 run api 1340000 1370000
wait
} | tee $O/r06_gpu_fuzz_b.txt
