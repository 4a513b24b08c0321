#!/bin/bash
# quick A/B: serial per-stage times of both scenes (no tests, no CPU baseline)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/quick.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/quick.json'))
print("d2   ", d['value'], d['config']['value_one_frame_at_a_time'], d['roofline']['stage_ms'])
s=d['config']['secondary']; print("r1mix", s['value'], s['value_one_frame_at_a_time'], s['roofline']['stage_ms'])
print(d['config']['bump'], s['bump'])
PY
