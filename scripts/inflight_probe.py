#!/usr/bin/env python3
"""Frames in flight N = 1..8 on one context (d2), a process per setting of GPU_MAX_HW_QUEUES: with the lanes' queues understood
(profiles/r06_queue_order.txt) is four still the best?   python scripts/inflight_probe.py [scene]"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vello_amd  # noqa: E402
from scripts.ab_process import workload  # noqa: E402

WHITE = 0xFFFFFFFF


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "d2"
    wl = workload(key)
    eng = vello_amd.Engine(capacities=wl.caps) if wl.caps else vello_amd.Engine()
    eng.upload_scene(wl.packed, wl.layout)
    w, h, aa = wl.width, wl.height, wl.aa
    ring = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(8)]
    torch.cuda.synchronize()
    for rep in range(2):
        for n in (1, 2, 3, 4, 5, 6, 8):
            eng.set_frames_in_flight(n)
            for i in range(2 * n + 8):
                eng.render_resident(w, h, WHITE, aa, out=ring[i % n])
            assert eng.sync() == 0
            t = time.perf_counter()
            for i in range(240):
                eng.render_resident(w, h, WHITE, aa, out=ring[i % n])
            assert eng.sync() == 0
            print(json.dumps({"scene": key, "hw_queues": os.environ["GPU_MAX_HW_QUEUES"], "in_flight": n, "rep": rep,
                              "fps": round(240 / (time.perf_counter() - t), 1)}), flush=True)


if __name__ == "__main__":
    main()
