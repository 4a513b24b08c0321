#!/usr/bin/env python3
"""Is the GPU fed fast enough?  rocprofv3's time line shows a lane idle for hundreds of microseconds between the end of its frame and
the first kernel of its next one (profiles/r06_pipeline_timeline_gaps.txt).  Measures, on one context with four lanes (d2):
  (a) the host's time to ENQUEUE a frame (render_resident on an idle lane returns when the launches are queued);
  (b) frames/s of the plain loop;
  (c) frames/s with N processes' worth of contexts driven by N host threads (ctypes calls release the GIL).
    python scripts/host_feed_probe.py [threads]"""
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vello_amd  # noqa: E402
from scripts.ab_process import workload  # noqa: E402

WHITE = 0xFFFFFFFF


def main():
    n_thr = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    wl = workload("d2")
    w, h, aa = wl.width, wl.height, wl.aa
    engs, rings = [], []
    for _ in range(n_thr):
        e = vello_amd.Engine(capacities=wl.caps)
        e.upload_scene(wl.packed, wl.layout)
        e.set_frames_in_flight(4)
        engs.append(e)
        rings.append([torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)])
    torch.cuda.synchronize()
    e, ring = engs[0], rings[0]
    for i in range(12):
        e.render_resident(w, h, WHITE, aa, out=ring[i % 4])
    e.sync()
    # (a) enqueue cost: four frames onto four idle lanes
    enq = []
    for rep in range(10):
        t = time.perf_counter()
        for i in range(4):
            e.render_resident(w, h, WHITE, aa, out=ring[i])
        enq.append((time.perf_counter() - t) / 4)
        e.sync()
    print(json.dumps({"enqueue_us_per_frame_idle_lanes": round(1e6 * sorted(enq)[len(enq) // 2], 1)}), flush=True)
    # (b) one thread, one context
    for rep in range(2):
        t = time.perf_counter()
        for i in range(300):
            e.render_resident(w, h, WHITE, aa, out=ring[i % 4])
        e.sync()
        print(json.dumps({"threads": 1, "fps": round(300 / (time.perf_counter() - t), 1)}), flush=True)

    # (c) n threads, a context each
    def work(k, n, out):
        eng, rg = engs[k], rings[k]
        for i in range(n):
            eng.render_resident(w, h, WHITE, aa, out=rg[i % 4])
        eng.sync()
        out[k] = time.perf_counter()

    for rep in range(2):
        out = [0.0] * n_thr
        ths = [threading.Thread(target=work, args=(k, 300, out)) for k in range(n_thr)]
        t = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        print(json.dumps({"threads": n_thr, "fps_total": round(n_thr * 300 / (max(out) - t), 1)}), flush=True)


if __name__ == "__main__":
    main()
