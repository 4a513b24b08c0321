#!/usr/bin/env python3
"""One build of the library in a process of its own (the context is the process's only one: its lanes get the same hardware queues
every time, profiles/r06_queue_order.txt): frames/s with four frames in flight, one-frame latency and the kernels one frame at a time.

    python scripts/ab_process.py LETTER [scene ...]     LETTER: A = the tree's library, X = ab_tmp/libvello_hip_X.so
    (alternate the letters from a shell loop: scripts/sessions/gpu_r6_s13.sh)"""
import json
import os
import pickle
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import vello_amd  # noqa: E402
import vello_amd._lib as L  # noqa: E402
from vello_amd.renderer import STAGES  # noqa: E402

WHITE = 0xFFFFFFFF
NIF = int(os.environ.get("VELLO_AB_NIF", "4"))  # frames in flight (the ring of targets is as deep)


def workload(key):
    """bench.Workload, cached across the processes of a session (generating d2 takes longer than measuring it)"""
    path = f"/tmp/vello_wl_{key}.pkl"
    if os.path.exists(path):
        with open(path, "rb") as f:
            return pickle.load(f)
    wl = bench.Workload(key, 0)
    wl.scene = None
    with open(path, "wb") as f:
        pickle.dump(wl, f)
    return wl


def main():
    letter = sys.argv[1]
    scenes = sys.argv[2:] or ["d2"]
    if letter != "A":
        L._use_library(os.path.join(ROOT, "ab_tmp", f"libvello_hip_{letter}.so"))
    for key in scenes:
        wl = workload(key)
        eng = vello_amd.Engine(capacities=wl.caps) if wl.caps else vello_amd.Engine()
        eng.upload_scene(wl.packed, wl.layout)
        w, h, aa = wl.width, wl.height, wl.aa
        ring = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(NIF)]
        torch.cuda.synchronize()
        eng.set_frames_in_flight(NIF)
        for i in range(20):
            eng.render_resident(w, h, WHITE, aa, out=ring[i % NIF])
        assert eng.sync() == 0, eng.bump()
        fps = []
        for _ in range(3):
            t = time.perf_counter()
            for i in range(200):
                eng.render_resident(w, h, WHITE, aa, out=ring[i % NIF])
            assert eng.sync() == 0
            fps.append(200 / (time.perf_counter() - t))
        eng.set_frames_in_flight(1)
        for _ in range(10):
            eng.render_resident(w, h, WHITE, aa, out=ring[0])
            eng.sync_frame(0)
        lat = []
        for _ in range(60):
            t = time.perf_counter()
            eng.render_resident(w, h, WHITE, aa, out=ring[0])
            eng.sync_frame(0)
            lat.append(time.perf_counter() - t)
        eng.set_profiling(STAGES)
        eng.stage_ms()
        eng.kernel_ms()
        for _ in range(30):
            eng.render_resident(w, h, WHITE, aa, out=ring[0])
            eng.sync_frame(0)
        st, km = eng.stage_ms(), eng.kernel_ms()
        eng.set_profiling([])
        k = {n[2:]: round(1e3 * v[0] / max(v[1], 1), 1) for n, v in km.items()}
        s = {n[:10]: round(1e3 * v[0] / max(v[1], 1), 1) for n, v in st.items() if n not in ("flatten", "coarse") and 1e3 * v[0] / max(v[1], 1) >= 5.0}
        print("%-6s %-2s %6.0f frames/s (%s)  %6.1f us | %s | %s" % (key, letter, statistics.median(fps), " ".join("%.0f" % f for f in fps),
              1e6 * statistics.median(lat), " ".join(f"{a} {b}" for a, b in k.items()), " ".join(f"{a} {b}" for a, b in s.items())), flush=True)
        del eng


if __name__ == "__main__":
    main()
