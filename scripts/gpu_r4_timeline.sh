#!/bin/bash
# kernel time line with frames in flight (rocprofv3 --kernel-trace only), summarised by scripts/pipeline_timeline.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4tl
for nif in ${NIF:-4}; do
  rm -rf /tmp/tl; mkdir -p /tmp/tl
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python bench.py --workload d2 --steps 40 --warmup 5 --in-flight $nif --timed-only > /dev/null 2>&1
  f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
  echo "== in flight $nif"; head -1 "$f" | cut -c1-300
  python scripts/pipeline_timeline.py "$f" | tee gpurun_out/r4tl/timeline_nif$nif.txt
done
