#!/bin/bash
# round 5, session 32: the kernel time line with four frames in flight / one and the frames-in-flight sweep, at the round's last kernel commit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s32
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
NIF="4 1" timeout 200 bash scripts/gpu_r4_timeline.sh > $O/r05_pipeline_timeline.txt 2>&1; grep -A3 "window" $O/r05_pipeline_timeline.txt | head -12
for nif in 2 3 4 5 6 8; do
  timeout 60 python bench.py --workload d2 --steps 200 --warmup 10 --no-cpu-baseline --timed-only --in-flight $nif 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight', $nif, d['value'])"
done > $O/r05_inflight_sweep.txt; cat $O/r05_inflight_sweep.txt
