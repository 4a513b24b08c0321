#!/bin/bash
# the bench line as the driver runs it (and with the defaults), saved under gpurun_out/r4bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4bench
t0=$(date +%s.%N); python bench.py --steps 20 --warmup 5 > gpurun_out/r4bench/bench_k20.json 2> gpurun_out/r4bench/bench_k20.err; echo "bench.py wall: $(echo "$(date +%s.%N) - $t0" | bc) s"
tail -3 gpurun_out/r4bench/bench_k20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4bench/bench_k20.json").read().strip().splitlines()[-1])
c = d["config"]; r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "one-frame", c["value_one_frame_at_a_time"], "200-step", c.get("value_200_steps"))
print("roofline", r["kernel"], r["avg_launch_ms"], r["frac"], "peak_measured", r["peak_measured"], "frame_frac", r["frame_frac"], r.get("frame_frac_without_tile_area"))
print("intervals", c["frame_completion_interval_ms_pipelined"], c.get("frame_completion_interval_ms_200_steps"))
for o in c.get("other_configs", []): print({k: o.get(k) for k in ("config", "value", "one_frame_latency_ms", "dominant_kernel", "dominant_kernel_ms", "frac", "error")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("value_single_thread"))
print("stage_ms", r["stage_ms"])
PY
