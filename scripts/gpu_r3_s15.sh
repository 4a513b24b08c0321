#!/bin/bash
# round 3 session 15: k_fine capped at 3 waves per SIMD by a dynamic-LDS pad (variant P, VELLO_FINE_LDS_PAD) -- does the room it leaves
# help the other frames' kernels when frames are in flight?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s15
mkdir -p $OUT
one() {
  python scripts/ab_bench.py P --steps 80 --warmup 10 --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['secondary']
f=lambda r: ' '.join('%s %.0f' % (k[:6], v*1e3) for k,v in r['stage_ms'].items() if v*1e3 >= 20)
print('$1 d2 %.0f/%.0f r1mix %.0f/%.0f | d2 [%s] r1mix [%s]' % (d['value'], d['config']['value_one_frame_at_a_time'], s['value'], s['value_one_frame_at_a_time'], f(d['roofline']), f(s['roofline'])))"
}
for rep in 1 2; do
  VELLO_FINE_LDS_PAD=0 one pad0 "" | tee -a $OUT/ab.txt
  VELLO_FINE_LDS_PAD=4096 one pad4096 "" | tee -a $OUT/ab.txt
  VELLO_FINE_LDS_PAD=1100 one pad1100 "" | tee -a $OUT/ab.txt
done
VELLO_FINE_LDS_PAD=4096 one pad4096_if6 "--in-flight 6" | tee -a $OUT/ab.txt
VELLO_FINE_LDS_PAD=0 one pad0_if6 "--in-flight 6" | tee -a $OUT/ab.txt
VELLO_FINE_LDS_PAD=4096 one pad4096_if8 "--in-flight 8" | tee -a $OUT/ab.txt
VELLO_FINE_LDS_PAD=0 one pad0_if8 "--in-flight 8" | tee -a $OUT/ab.txt
