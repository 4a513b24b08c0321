#!/bin/bash
# The first GPU session after round 2 (about 3 minutes): where k_fine's cycles go, then same-box A/B of the in-tree build (A)
# against the patched builds (B merged restore, C one LDS entry per staged fill, D graded priority, E path_count chunks of 2048 lines) and against the
# two-waves-per-tile kernel (P = the in-tree library with VELLO_FINE_PIPELINE=1).  Everything under `timeout`: k_fine_pipe has
# never run on hardware.        scripts/prepare_experiments.sh && scripts/grun.sh --timeout 300 -- 'bash scripts/gpu_experiments.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/experiments
mkdir -p $OUT
timeout 60 python scripts/fine_prof.py d2 r1mix > $OUT/fine_prof.txt 2>&1; tail -40 $OUT/fine_prof.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['config']['secondary']
print('$1 d2 %.0f/%.0f r1mix %.0f/%.0f | fine %.0f / %.0f us' % (d['value'], d['config']['value_one_frame_at_a_time'], s['value'], s['value_one_frame_at_a_time'], d['roofline']['stage_ms']['fine']*1e3, s['roofline']['stage_ms']['fine']*1e3))"; }
# the pipeline kernel first on a small frame: does it run at all, and is its image the one-wave kernel's
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
  for w in A B C D E; do
    timeout 60 python scripts/ab_bench.py $w --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | line $w | tee -a $OUT/ab.txt
  done
  VELLO_FINE_PIPELINE=1 timeout 60 python scripts/ab_bench.py A --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | line P | tee -a $OUT/ab.txt
done
