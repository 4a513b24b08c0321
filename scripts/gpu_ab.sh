#!/bin/bash
# same-box A/B: in-tree build (A) vs ab_tmp/libvello_hip_<X>.so for X in $VARIANTS (default B), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for rep in ${REPS:-1 2}; do
for w in A ${VARIANTS:-B}; do
  python scripts/ab_bench.py $w --steps ${STEPS:-80} --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['secondary']
f=lambda r: ' '.join('%s %.0f' % (k[:6], v*1e3) for k,v in r['stage_ms'].items() if v*1e3 >= 20)
print('$w d2 %.0f/%.0f r1mix %.0f/%.0f | d2 [%s] r1mix [%s]' % (d['value'], d['config']['value_one_frame_at_a_time'], s['value'], s['value_one_frame_at_a_time'], f(d['roofline']), f(s['roofline'])))"
done
done
