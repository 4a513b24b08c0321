#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV: which hardware queue (Queue_Id) every HIP stream (Stream_Id) dispatched on, in order of
first appearance, with the number of k_fine launches (a lane's frames).  For scripts/stream_order_probe.py."""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
streams = OrderedDict()
for r in rows:
    s = streams.setdefault(r["Stream_Id"], {"queues": OrderedDict(), "fine": 0, "first": int(r["Start_Timestamp"])})
    s["queues"][r["Queue_Id"]] = s["queues"].get(r["Queue_Id"], 0) + 1
    if "k_fine" in r["Kernel_Name"]:
        s["fine"] += 1
t0 = min(s["first"] for s in streams.values())
for sid, s in streams.items():
    print(f"stream {sid:>3s}  first use +{(s['first'] - t0) / 1e6:9.1f} ms  k_fine launches {s['fine']:5d}  queues {dict(s['queues'])}")
