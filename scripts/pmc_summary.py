"""Condenses a rocprofv3 counter_collection.csv into per-kernel averages per dispatch."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in rows:
    k = r.get("Kernel_Name", "?").split("(")[0]
    c = r.get("Counter_Name")
    v = float(r.get("Counter_Value", 0) or 0)
    acc[k][c] += v
    cnt[k][c] += 1
for k in sorted(acc):
    if not k.startswith(("vk::", "void vk::", "calib_")):
        continue
    print(k)
    for c in sorted(acc[k]):
        print(f"    {c:28s} avg/dispatch {acc[k][c] / max(cnt[k][c], 1):16.1f}   (n={cnt[k][c]})")
