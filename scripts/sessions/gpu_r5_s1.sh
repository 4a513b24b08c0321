#!/bin/bash
# round 5, session 1: measurements only (nothing built yet).  The new hygiene tests, flatten's phase profile on mmark / tiger / d2
# (VERDICT r4 item 2), the brush specialisation's stages and phases (item 6), the stage-synchronous batching estimate (item 3a),
# and a bench line of the tree as round 4 left it on this box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s1
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py -x -q -k "primitive or handoff_stress or native_library" 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python scripts/flatten_prof.py mmark tiger d2 2>&1 | grep -v amdgpu.ids | tee $O/flatten_prof.txt
timeout 200 python scripts/brush_prof.py stages 2>&1 | grep -v amdgpu.ids | tee $O/brush_stages.jsonl
timeout 200 python scripts/brush_prof.py phases 2>&1 | grep -v amdgpu.ids | tee $O/brush_phases.txt
timeout 400 python scripts/batch_estimate.py 2>&1 | grep -v amdgpu.ids | tee $O/batch_estimate.txt
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5s1/bench.json").read().strip().splitlines()[-1])
c = d["config"]; r = d["roofline"]
print("value", d["value"], "one-frame", c["value_one_frame_at_a_time"], "fine", r["avg_launch_ms"], r["frac"])
print("stage_ms", r["stage_ms"])
for o in c.get("other_configs", []): print({k: o.get(k) for k in ("config", "value", "one_frame_latency_ms", "dominant_kernel", "dominant_kernel_ms")})
PY
