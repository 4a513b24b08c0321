#!/bin/bash
# frames in flight x hardware queues with the round's kernels (d2 and r1mix, 200 timed steps each)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4sweep
for wl in d2 r1mix; do for q in 8; do for nif in 2 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --timed-only --in-flight $nif 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl queues', $q, 'in flight', $nif, d['value'])"
done; done; done | tee gpurun_out/r4sweep/sweep.txt
