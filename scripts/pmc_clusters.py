"""Per-kernel counter averages of a rocprofv3 counter_collection.csv, split into the dispatches above and below the kernel's mean
(ab_bench.py renders two scenes in turn: the d2 frames and the r1mix frames of one kernel come out as the two clusters)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else "k_path_count"
vals = collections.defaultdict(list)
for r in rows:
    k = r.get("Kernel_Name", "?").split("(")[0]
    if want in k:
        vals[(k, r["Counter_Name"])].append(float(r.get("Counter_Value", 0) or 0))
for (k, c), v in sorted(vals.items()):
    m = sum(v) / len(v)
    hi = [x for x in v if x >= m] or [0.0]
    lo = [x for x in v if x < m] or [0.0]
    print(f"{k:36s} {c:24s} high {sum(hi) / len(hi):14.0f} (n={len(hi)})   low {sum(lo) / len(lo):14.0f} (n={len(lo)})")
