#!/usr/bin/env python3
"""Why do contexts of ONE build, created one after the other in ONE process, differ by up to 13 % in frames/s with four frames in flight
(profiles/r06_ab_path_count_footprint3.txt: H0 2120, H3 2394)?  Candidates: which hardware queues the lanes' streams land on (HIP
maps streams to GPU_MAX_HW_QUEUES queues in creation order), or where the pools were allocated.

    python scripts/stream_order_probe.py [n_contexts] [dummy streams before context k: comma list]

Creates contexts one after the other, each preceded by its number of dummy streams (hipStreamCreateWithFlags, never used), measures
each context right after its creation and all of them again at the end."""
import ctypes
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

WHITE = 0xFFFFFFFF
hip = ctypes.CDLL("libamdhip64.so")


def dummy_streams(n):
    out = []
    for _ in range(n):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
        out.append(s)
    return out


def fps(eng, wl, ring, frames=100):
    w, h, aa = wl.width, wl.height, wl.aa
    eng.set_frames_in_flight(4)
    for i in range(12):
        eng.render_resident(w, h, WHITE, aa, out=ring[i % 4])
    assert eng.sync() == 0
    out = []
    for _ in range(2):
        t = time.perf_counter()
        for i in range(frames):
            eng.render_resident(w, h, WHITE, aa, out=ring[i % 4])
        assert eng.sync() == 0
        out.append(round(frames / (time.perf_counter() - t), 1))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dummies = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0] * n
    dummies += [0] * (n - len(dummies))
    wl = bench.Workload("d2", 0)
    ring = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
    torch.cuda.synchronize()
    keep, engines = [], []
    for k in range(n):
        keep += dummy_streams(dummies[k])
        e = vello_amd.Engine(capacities=wl.caps)
        e.upload_scene(wl.packed, wl.layout)
        engines.append(e)
        print(json.dumps({"context": k, "dummy_streams_before": dummies[k], "when": "created", "fps": fps(e, wl, ring)}), flush=True)
    for k, e in enumerate(engines):
        print(json.dumps({"context": k, "when": "end", "fps": fps(e, wl, ring)}), flush=True)


if __name__ == "__main__":
    main()
