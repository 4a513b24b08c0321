#!/bin/bash
# Round 6 session 45: the round's last tree (kernels of edb47a6, the evidence session's pmc_traffic.json in place): the GPU suite and the bench line with the driver's flags
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s45
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/r06_gputest.log; tail -2 $OUT/r06_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/r06_bench_driver_flags_k20.json; head -c 200 $OUT/r06_bench_driver_flags_k20.json; echo
python -c "
import json; d=json.loads(open('$OUT/r06_bench_driver_flags_k20.json').read()); print(d['value'], d['config'].get('value_200_steps'), d['roofline']['traffic_stale'], d['roofline']['frac'])"
