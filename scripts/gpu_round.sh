#!/bin/bash
# One GPU session: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > gpurun_out/rocminfo.txt
nproc > gpurun_out/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/nproc.txt
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py --steps ${BENCH_STEPS:-100} --warmup 10 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/rocprof.log 2>&1
ls gpurun_out/prof 2>/dev/null | head
find gpurun_out/prof -name "*kernel_stats*" | head -2 | xargs -r head -30
rm -f gpurun_out/prof/*kernel_trace.csv gpurun_out/prof/*.db
