#!/bin/bash
# Round 3, first GPU session: the GPU suite at HEAD (twice over the image tests: the atlas race of ADVICE r2 was timing dependent),
# k_fine's phase timers, and the same-box A/B of round 2's waiting experiments.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s1
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $OUT/gputest.log; tail -3 $OUT/gputest.log
for i in 1 2 3 4; do (timeout 120 python -m pytest tests -m gpu -q -k "brushes or atlas or image" 2>&1 | tail -2) >> $OUT/gputest_images.log; done; tail -4 $OUT/gputest_images.log
timeout 90 python scripts/fine_prof.py d2 r1mix > $OUT/fine_prof.txt 2>&1; tail -44 $OUT/fine_prof.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['secondary']
f=lambda r: ' '.join('%s %.0f' % (k[:6], v*1e3) for k,v in r['stage_ms'].items() if v*1e3 >= 20)
print('$1 d2 %.0f/%.0f r1mix %.0f/%.0f | d2 [%s] r1mix [%s]' % (d['value'], d['config']['value_one_frame_at_a_time'], s['value'], s['value_one_frame_at_a_time'], f(d['roofline']), f(s['roofline'])))"; }
timeout 60 python - <<'PY' 2>&1 | tail -3
import numpy as np, vello_amd, workloads
from vello_amd import AaConfig
eng = vello_amd.Engine(device=0)
p, l = workloads.random_test_scene(3, n_paths=600, size=512.0, strokes=True, clips=True).resolve()
for aa in (AaConfig.Msaa8, AaConfig.Msaa16):
    eng.set_debug_flags(); a, _ = eng.render(p, l, 512, 512, 0xff000000, aa)
    eng.set_debug_flags(fine_pipeline=True); b, _ = eng.render(p, l, 512, 512, 0xff000000, aa)
    print("k_fine_pipe", int(aa), "image equal to k_fine's:", bool(np.array_equal(a, b)))
PY
for rep in 1 2; do
  for w in A B D E; do
    timeout 90 python scripts/ab_bench.py $w --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | line $w | tee -a $OUT/ab.txt
  done
  VELLO_FINE_PIPELINE=1 timeout 90 python scripts/ab_bench.py A --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | line P | tee -a $OUT/ab.txt
done
