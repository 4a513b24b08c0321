#!/bin/bash
# per-kernel durations, one frame at a time (rocprofv3 --kernel-trace --stats), for WL in d2 r1mix ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for wl in ${WL:-d2}; do
  rm -rf /tmp/ks; mkdir -p /tmp/ks
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python bench.py --workload $wl --steps 20 --warmup 3 --in-flight 1 --timed-only > /dev/null 2>&1
  f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
  echo "== $wl"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f'{r["Name"][:48]:48s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
done
