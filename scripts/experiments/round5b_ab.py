"""Round 5, second sitting: the same-box A/B of its three steps, in ONE process (one torch import, every scene made once).

    python scripts/round5b_ab.py [reps]        -> lines of JSON on stdout, a table on stderr

Variants, alternating rep by rep on d2, r1mix, the tiger (1024^2 MSAA8) and mmark-50k (2048^2 MSAA16):
    P    ab_tmp/libvello_hip_P.so: the library at 181780a (before this sitting)
    C    the tree's library with VELLO_HIP_DEBUG_NO_PREZERO: coarse's tile bits a word per 8 tiles, nothing else
    A    the tree's library as shipped: + the tiles of a finished frame zeroed beside k_flatten_light
Per variant: frames/s with four frames in flight (2 x 100 frames), the median latency of 60 frames rendered one at a time, and the
stages' times one frame at a time (HIP events around every stage, 30 frames).
Then, on d2 only, lane streams on partitions of the CUs (VELLO_HIP_LANE_CU_SPLIT, read when a lane's stream is created): frames/s with
four frames in flight for 2 / 2i / 4 / 4i against none."""
import json
import os
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import vello_amd  # noqa: E402
import vello_amd._lib as L  # noqa: E402
from vello_amd.renderer import STAGES  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
WHITE = 0xFFFFFFFF
# R5B_DRY=1: the script's own logic, without a GPU -- every variant is the SIMT-emulator build, the scenes are small, the frames land in
# the engine's own output buffer
DRY = os.environ.get("R5B_DRY") == "1"
SCENES = ("d2", "r1mix", "tiger", "mmark") if not DRY else ("d2", "tiger")


def n_frames(n):  # (the dry run renders a few frames where the real one renders a hundred)
    return max(2, n // 30) if DRY else n


class DryWorkload:
    def __init__(self, key, rank):
        import workloads
        self.key, self.caps = key, None
        self.width = self.height = 128
        self.aa = vello_amd.AaConfig.Msaa16
        self.packed, self.layout = workloads.random_test_scene(3, n_paths=40, size=128.0, strokes=True, clips=False).resolve()


def make_workload(key):
    return DryWorkload(key, 0) if DRY else bench.Workload(key, 0)


def make_ring(wl, n=4):
    if DRY:
        return [None] * n
    ring = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(n)]
    torch.cuda.synchronize()
    return ring


def frame_of(eng, wl, ring):
    eng.render_resident(wl.width, wl.height, WHITE, wl.aa, out=ring[0])
    assert eng.sync() == 0
    if DRY:
        import numpy as np
        return torch.from_numpy(eng.read_buffer("output", np.uint8, wl.width * wl.height * 4).copy())
    return ring[0].clone()


def make_engine(variant, caps):
    if DRY:
        L._use_library(os.path.join(ROOT, "tests", "simt_emu", "libvello_emu.so"))
    elif variant == "P":
        L._use_library(os.path.join(ROOT, "ab_tmp", "libvello_hip_P.so"))
    try:
        eng = vello_amd.Engine(capacities=caps) if caps else vello_amd.Engine()
    finally:
        L._use_library(None)
    if variant == "C":
        eng.set_debug_flags(no_prezero=True)
    return eng


def measure(eng, wl, ring):
    w, h, aa = wl.width, wl.height, wl.aa
    nif = len(ring)
    eng.set_frames_in_flight(nif)
    for i in range(n_frames(12)):
        eng.render_resident(w, h, WHITE, aa, out=ring[i % nif])
    assert eng.sync() == 0, eng.bump()
    fps = []
    for _ in range(2):
        t = time.perf_counter()
        for i in range(n_frames(100)):
            eng.render_resident(w, h, WHITE, aa, out=ring[i % nif])
        assert eng.sync() == 0
        fps.append(n_frames(100) / (time.perf_counter() - t))
    eng.set_frames_in_flight(1)
    for _ in range(n_frames(10)):
        eng.render_resident(w, h, WHITE, aa, out=ring[0])
        eng.sync_frame(0)
    lat = []
    for _ in range(n_frames(60)):
        t = time.perf_counter()
        eng.render_resident(w, h, WHITE, aa, out=ring[0])
        eng.sync_frame(0)
        lat.append(time.perf_counter() - t)
    eng.set_profiling(STAGES)
    eng.stage_ms()
    eng.kernel_ms()
    for _ in range(n_frames(30)):
        eng.render_resident(w, h, WHITE, aa, out=ring[0])
        eng.sync_frame(0)
    st = eng.stage_ms()
    km = eng.kernel_ms()
    eng.set_profiling([])
    stage_us = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in st.items()}
    kernel_us = {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in km.items()}
    return {"fps_4_in_flight": [round(f, 1) for f in fps], "latency_us": round(1e6 * statistics.median(lat), 1), "stage_us": stage_us,
            "kernel_us": kernel_us, "prezero_tiles": eng.last_prezero_tiles() if hasattr(eng._lib, "vello_hip_last_prezero_tiles") else None}


def main():
    results = {}
    for key in SCENES:
        wl = make_workload(key)
        ring = make_ring(wl)
        engines = {}
        for v in ("P", "C", "A"):
            engines[v] = make_engine(v, wl.caps)
            engines[v].upload_scene(wl.packed, wl.layout)
        for rep in range(REPS):
            for v in ("P", "C", "A"):
                r = measure(engines[v], wl, ring)
                r.update({"scene": key, "variant": v, "rep": rep})
                print(json.dumps(r), flush=True)
                results.setdefault((key, v), []).append(r)
        # the three variants must show the same frame
        imgs = {v: frame_of(engines[v], wl, ring) for v in ("P", "C", "A")}
        same = bool(torch.equal(imgs["P"], imgs["A"]) and torch.equal(imgs["P"], imgs["C"]))
        print(json.dumps({"scene": key, "images_equal": same}), flush=True)
        del engines
    # summary
    for (key, v), rs in sorted(results.items()):
        fps = statistics.median([f for r in rs for f in r["fps_4_in_flight"]])
        lat = statistics.median([r["latency_us"] for r in rs])
        st = {k: statistics.median([r["stage_us"][k] for r in rs]) for k in rs[0]["stage_us"]}
        sys.stderr.write("%-6s %s  %7.0f frames/s  %7.1f us  | %s\n" % (key, v, fps, lat, " ".join("%s %.0f" % (k[:9], x) for k, x in st.items() if x >= 8)))
    # lane streams on partitions of the CUs (d2, four frames in flight)
    wl = make_workload("d2")
    ring = make_ring(wl)
    for split in ("", "2", "2i", "4", "4i", ""):
        if split:
            os.environ["VELLO_HIP_LANE_CU_SPLIT"] = split
        else:
            os.environ.pop("VELLO_HIP_LANE_CU_SPLIT", None)
        try:
            eng = make_engine("A", wl.caps)
            eng.upload_scene(wl.packed, wl.layout)
            eng.set_frames_in_flight(4)
            for i in range(n_frames(16)):
                eng.render_resident(wl.width, wl.height, WHITE, wl.aa, out=ring[i % 4])
            assert eng.sync() == 0
            fps = []
            for _ in range(3):
                t = time.perf_counter()
                for i in range(n_frames(100)):
                    eng.render_resident(wl.width, wl.height, WHITE, wl.aa, out=ring[i % 4])
                assert eng.sync() == 0
                fps.append(round(n_frames(100) / (time.perf_counter() - t), 1))
            print(json.dumps({"scene": "d2", "lane_cu_split": split or "none", "fps_4_in_flight": fps}), flush=True)
            sys.stderr.write("d2 lane CU split %-5s %s\n" % (split or "none", fps))
            del eng
        except Exception as e:  # (a mask the runtime refuses must not cost the session its other results)
            print(json.dumps({"scene": "d2", "lane_cu_split": split, "error": str(e)}), flush=True)
    os.environ.pop("VELLO_HIP_LANE_CU_SPLIT", None)


if __name__ == "__main__":
    main()
