#!/bin/bash
# round 5, session 33: rocprofv3 kernel statistics of BASELINE's other single-GPU configs (C2 tiger 1024^2 MSAA8, C4 mmark-50k 2048^2 MSAA16), one
# frame at a time, at the round's last kernel commit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s33
mkdir -p $O
cp .commit_stamp $O/commit.txt 2>/dev/null || true
for wl in tiger mmark; do
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$wl -o p -- python scripts/render_loop.py $wl 60 > $O/$wl.log 2>&1
  find $O/tmp_$wl -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $O/r05_kernel_stats_serial_$wl.csv
  rm -rf $O/tmp_$wl; tail -1 $O/$wl.log; head -8 $O/r05_kernel_stats_serial_$wl.csv | cut -c1-120
done
