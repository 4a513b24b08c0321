cd "${GRAFT_REPO_ROOT:-/root/repo}"
for q in 4 8 16 24; do for nif in 4 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --workload d2 --steps 200 --warmup 10 --no-cpu-baseline --timed-only --in-flight $nif 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues', $q, 'inflight', $nif, d['value'])"
done; done
