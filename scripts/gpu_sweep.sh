#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
bash scripts/gpu_quick.sh
for nif in 2 3 4 6 8; do
  python bench.py --workload d2 --steps 120 --warmup 10 --no-cpu-baseline --timed-only --in-flight $nif 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflight', $nif, d['value'])"
done
