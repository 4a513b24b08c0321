#!/bin/bash
# Round 6 session 46: the stroke kernel hands non-finite / overflowing inputs to the heavy code (the NaN-box findings of the forced kernel: seeds 4552, 8707, 11797, 11851):
# GPU suite, tree (A) against HEAD's library (H), and the extreme-value fuzz with the kernel forced over the ranges that held the findings + new ones
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s46
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gputest.log; tail -2 $OUT/gputest.log
for rep in 1 2 3; do for L in H A; do timeout 120 python scripts/ab_process.py $L d2 mmark 2>/dev/null | cut -c1-270; done; done > $OUT/ab_stroke_tame.txt
cat $OUT/ab_stroke_tame.txt
export FUZZ_GPU=1
T=${T:-300}
n=0
run() { n=$((n+1)); ( timeout $T python scripts/fuzz_campaign.py "$@" 2>&1 | grep -E "SEED|done|^at " | tail -3 | sed "s/^/[$n: ${FUZZ_FLATTEN:-auto} sk=${FUZZ_STROKE_KERNEL:-0} nif=${FUZZ_IN_FLIGHT:-1} $*] /" ) & }
{
FUZZ_STROKE_KERNEL=1 run extreme 4300 4800
FUZZ_STROKE_KERNEL=1 run extreme 8500 9000
FUZZ_STROKE_KERNEL=1 run extreme 11700 12100
FUZZ_STROKE_KERNEL=1 run extreme 20000 22000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=4 run extreme 22000 24000
FUZZ_STROKE_KERNEL=1 FUZZ_IN_FLIGHT=2 run api 5000000 5100000
FUZZ_STROKE_KERNEL=1 FUZZ_FLATTEN=coop run sizes 700000 720000
run extreme 24000 26000
wait
} | tee $OUT/fuzz_stroke_tame.txt
