#!/bin/bash
# Calibrates FETCH_SIZE / WRITE_SIZE on known byte counts (scripts/calib/pmc_calib.hip), then collects the same
# two counters for the engine's kernels.  Separate --pmc passes, kernel trace only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { # name counter cmd...
  name=$1; ctr=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc/$name -o p -- "$@" > gpurun_out/pmc/$name.log 2>&1
  f=$(find gpurun_out/pmc/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > gpurun_out/pmc/$name.summary.txt; cat gpurun_out/pmc/$name.summary.txt; rm -rf gpurun_out/pmc/$name; else tail -5 gpurun_out/pmc/$name.log; fi
}
run calib_fetch FETCH_SIZE scripts/calib/pmc_calib
run calib_write WRITE_SIZE scripts/calib/pmc_calib
run eng_fetch FETCH_SIZE python bench.py --steps 10 --warmup 2 --in-flight 1 --timed-only
run eng_write WRITE_SIZE python bench.py --steps 10 --warmup 2 --in-flight 1 --timed-only
