#!/bin/bash
# round 5, session 17: fewer waves of k_fine per CU (dynamic LDS on top of its own: VELLO_HIP_FINE_LDS_PAD) -- does leaving room for the
# other frames' kernels raise the frame rate with four frames in flight?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s17
mkdir -p $O
for rep in 1 2; do for pad in 0 1200 3200 6400 10400; do for nif in 4 1; do
  VELLO_HIP_FINE_LDS_PAD=$pad python bench.py --workload d2 --steps 200 --warmup 10 --no-cpu-baseline --timed-only --in-flight $nif 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad', $pad, 'in flight', $nif, d['value'])"
done; done; done | tee $O/fine_lds_pad.txt
