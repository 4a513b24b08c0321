#!/usr/bin/env python3
"""bench.py's timed loop reaches 2170 frames/s on d2 where a bare render_resident loop on the same context reaches 2340
(scripts/stream_order_probe.py): which part of the loop costs it?  One context, the variants measured alternately.

    python scripts/bench_loop_probe.py"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import vello_amd  # noqa: E402
from vello_amd.distributed import FramePipeline  # noqa: E402

WHITE = 0xFFFFFFFF


def main():
    if os.environ.get("PROBE_LIBRARY"):  # another build of the library (ab_tmp/libvello_hip_<X>.so)
        import vello_amd._lib as L
        L._use_library(os.path.join(ROOT, os.environ["PROBE_LIBRARY"]))
    wl = bench.Workload("d2", 0)
    mode = sys.argv[1] if len(sys.argv) > 1 else "ring_last"
    if mode == "tiny_first":  # a kernel on torch's stream (the null stream) before the engine exists; the ring last
        _t = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
    if mode == "empty_first":  # an allocation of the ring's size before the engine exists, no kernel; the ring last
        _t = torch.empty((4, wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0")
    if mode == "ring_first":
        ring = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
        torch.cuda.synchronize()
    eng = vello_amd.Engine(capacities=wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    if mode == "ring_mid":
        ring = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
        torch.cuda.synchronize()
    eng.set_frames_in_flight(4)
    if mode in ("ring_last", "tiny_first", "empty_first"):
        ring = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
        torch.cuda.synchronize()
    w, h, aa = wl.width, wl.height, wl.aa

    def bare(n):
        for i in range(n):
            eng.render_resident(w, h, WHITE, aa, out=ring[i % 4])
        assert eng.sync() == 0

    def piped(n):
        pipe = FramePipeline(4, render=lambda s: eng.render_resident(w, h, WHITE, aa, out=ring[s]), wait_frame=eng.sync_frame,
                             exchange=lambda s: None, wait_exchange=lambda ev: None)
        for _ in range(n):
            pipe.step()
        pipe.flush()
        assert eng.sync() == 0

    variants = [("bare", bare, []), ("bare+events(fine)", bare, ["fine"]), ("pipeline", piped, []), ("pipeline+events(fine)", piped, ["fine"])]
    if mode != "ring_last":
        variants = variants[:1]
    bare(20)
    for rep in range(3):
        for name, fn, prof in variants:
            eng.set_profiling(prof)
            fn(8)
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn(200)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            eng.stage_ms(); eng.kernel_ms()
            eng.set_profiling([])
            print(json.dumps({"mode": mode, "variant": name, "rep": rep, "fps": round(200 / dt, 1)}), flush=True)


if __name__ == "__main__":
    main()
