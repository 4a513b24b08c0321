#!/bin/bash
# Round 5 evidence for one commit, one GPU session: the GPU suite, the bench line as the driver runs it (--steps 20 --warmup 5) and with
# the defaults (CPU baseline included), rocprofv3 kernel stats / PMC traffic / SQ counters (scripts/gpu_profiles.sh), the kernel time
# line with frames in flight, the other workloads, the brush kernels' stages and phases, flatten's and fine's phase profiles (a
# measurement build ab_tmp/libvello_hip_PROF.so must travel along: PROF_FLAGS="-DVELLO_FINE_PROF -DVELLO_COARSE_PROF
# -DVELLO_FLATTEN_PROF" bash scripts/build_prof.sh), the frames-in-flight sweep.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_evidence
mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null || true
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/r05_gputest.log; tail -2 $OUT/r05_gputest.log
timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/r05_bench_driver_flags_k20.json; head -c 300 $OUT/r05_bench_driver_flags_k20.json; echo
timeout 400 python bench.py 2>/dev/null | tail -1 > $OUT/r05_bench.json; head -c 300 $OUT/r05_bench.json; echo
TAG=r05 bash scripts/gpu_profiles.sh > $OUT/profiles.log 2>&1; tail -3 $OUT/profiles.log
NIF="4 1" bash scripts/gpu_r4_timeline.sh > $OUT/r05_pipeline_timeline.txt 2>&1; grep -A3 "window" $OUT/r05_pipeline_timeline.txt | head -12
timeout 300 python scripts/other_workloads.py 2>/dev/null > $OUT/r05_other_workloads.jsonl; wc -l $OUT/r05_other_workloads.jsonl
(timeout 200 python scripts/brush_prof.py stages 2>&1; timeout 200 python scripts/brush_prof.py phases 2>&1) | grep -v amdgpu.ids > $OUT/r05_brush_prof.txt
timeout 300 python scripts/flatten_prof.py mmark tiger d2 r1mix 2>&1 | grep -v amdgpu.ids > $OUT/r05_flatten_prof_end.txt
timeout 300 python scripts/fine_prof.py d2 r1mix 2>&1 | grep -v amdgpu.ids > $OUT/r05_fine_prof_end.txt
timeout 300 python scripts/flatten_kernels.py A 2>/dev/null | grep -v amdgpu.ids > $OUT/r05_flatten_kernels.txt
for nif in 2 3 4 5 6 8; do
  python bench.py --workload d2 --steps 200 --warmup 10 --no-cpu-baseline --timed-only --in-flight $nif 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight', $nif, d['value'])"
done > $OUT/r05_inflight_sweep.txt; cat $OUT/r05_inflight_sweep.txt
