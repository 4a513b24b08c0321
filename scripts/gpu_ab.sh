#!/bin/bash
# same-box A/B: in-tree build (A) vs ab_tmp/libvello_hip_B.so (B), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for rep in 1 2; do
for w in A B; do
  python scripts/ab_bench.py $w --steps ${STEPS:-80} --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['secondary']
print('$w d2 %.0f/%.0f fine %.1f coarse %.1f flat %.1f | r1mix %.0f/%.0f fine %.1f coarse %.1f flat %.1f' % (d['value'], d['config']['value_one_frame_at_a_time'], d['roofline']['stage_ms']['fine']*1e3, d['roofline']['stage_ms']['coarse']*1e3, d['roofline']['stage_ms']['flatten']*1e3, s['value'], s['value_one_frame_at_a_time'], s['roofline']['stage_ms']['fine']*1e3, s['roofline']['stage_ms']['coarse']*1e3, s['roofline']['stage_ms']['flatten']*1e3))"
done
done
