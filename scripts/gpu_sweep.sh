#!/bin/bash
# frames-in-flight sweep (pipelined throughput), both workloads, two passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for rep in 1 2; do
for wl in d2 r1mix; do
for nif in ${NIF:-2 3 4 5 6 8}; do
  python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --timed-only --in-flight $nif 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl inflight', $nif, d['value'])"
done
done
done
