#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > gpurun_out/rocminfo.txt
nproc > gpurun_out/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/nproc.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest -m gpu"
timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
fi
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-200} --warmup 20 2>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
echo "== rocprof (one frame at a time, d2 + r1mix)"
for wl in d2 r1mix; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$wl -o bench -- python bench.py --workload $wl --steps 40 --warmup 5 --in-flight 1 --timed-only > gpurun_out/rocprof_$wl.log 2>&1
find gpurun_out/prof_$wl -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} gpurun_out/kernel_stats_serial_$wl.csv
head -16 gpurun_out/kernel_stats_serial_$wl.csv
rm -rf gpurun_out/prof_$wl
done
fi
