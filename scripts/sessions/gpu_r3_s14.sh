#!/bin/bash
# round 3 session 14: flatten side by side with one frame in flight, one after the other with several (A) against HEAD (H)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s14
mkdir -p $OUT
one() {
  python scripts/ab_bench.py $2 --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['secondary']
f=lambda r: ' '.join('%s %.0f' % (k[:6], v*1e3) for k,v in r['stage_ms'].items() if v*1e3 >= 20)
k=lambda r: ' '.join('%s %.0f' % (k[2:], v*1e3) for k,v in r.get('kernel_ms_of_multi_kernel_stages', {}).items())
print('$1 d2 %.0f/%.0f r1mix %.0f/%.0f | d2 [%s] r1mix [%s] | d2 k [%s]' % (d['value'], d['config']['value_one_frame_at_a_time'], s['value'], s['value_one_frame_at_a_time'], f(d['roofline']), f(s['roofline']), k(d['roofline'])))"
}
(timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $OUT/gputest.log; tail -2 $OUT/gputest.log
for rep in 1 2 3; do
  one A A | tee -a $OUT/ab.txt
  one H H | tee -a $OUT/ab.txt
done
for w in A H; do python scripts/stage_times.py $w 2>/dev/null | tee $OUT/stages_$w.txt; done
