#!/bin/bash
# round 3 session 17: draw stage's workgroups FIRST in k_flatten_light's grid (A), PATH-marker fields written by the scan but the draw stage a
# launch of its own (N), the commit before (H)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s17
mkdir -p $OUT
one() {
  python scripts/ab_bench.py $2 --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['secondary']
f=lambda r: ' '.join('%s %.0f' % (k[:6], v*1e3) for k,v in r['stage_ms'].items() if v*1e3 >= 20)
print('$1 d2 %.0f/%.0f r1mix %.0f/%.0f | d2 [%s] r1mix [%s]' % (d['value'], d['config']['value_one_frame_at_a_time'], s['value'], s['value_one_frame_at_a_time'], f(d['roofline']), f(s['roofline'])))"
}
for rep in 1 2 3; do
  one A A | tee -a $OUT/ab.txt
  one N N | tee -a $OUT/ab.txt
  one H H | tee -a $OUT/ab.txt
done
