"""Same-box A/B of builds of libvello_hip.so in ONE process: a context per letter of ORDER (A = the tree's library, any other letter X =
ab_tmp/libvello_hip_X.so), created in that order and measured alternately, rep by rep -- per scene frames/s with four frames in flight
(2 x 100 frames), the median latency of 60 frames rendered one at a time, every stage's and kernel's time one frame at a time (HIP
events, 30 frames), and whether all contexts show the same image.  Repeat a letter (e.g. AQAQ) to see what the order of creation does
by itself.  One torch import and one scene generation for everything: a session of a few seconds.

    python scripts/ab_contexts.py ORDER [reps] [scene ...]        scenes: d2 r1mix tiger mmark (default: all four)
    AB_FLAGS_<letter>=flag,flag   debug flags of that letter's contexts (Engine.set_debug_flags names), e.g. AB_FLAGS_A=coarse_split"""
import json
import os
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


def make_engine(letter, caps):
    if letter != "A":
        L._use_library(os.path.join(ROOT, "ab_tmp", f"libvello_hip_{letter}.so"))
    try:
        eng = vello_amd.Engine(capacities=caps) if caps else vello_amd.Engine()
    finally:
        L._use_library(None)
    flags = [f for f in os.environ.get("AB_FLAGS_" + letter, "").split(",") if f]
    if flags:
        eng.set_debug_flags(**{f: True for f in flags})
    return eng


def measure(eng, wl, ring):
    w, h, aa = wl.width, wl.height, wl.aa
    nif = len(ring)
    eng.set_frames_in_flight(nif)
    for i in range(12):
        eng.render_resident(w, h, WHITE, aa, out=ring[i % nif])
    assert eng.sync() == 0, eng.bump()
    fps = []
    for _ in range(2):
        t = time.perf_counter()
        for i in range(100):
            eng.render_resident(w, h, WHITE, aa, out=ring[i % nif])
        assert eng.sync() == 0
        fps.append(100 / (time.perf_counter() - t))
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
    return {"fps_4_in_flight": [round(f, 1) for f in fps], "latency_us": round(1e6 * statistics.median(lat), 1),
            "stage_us": {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in st.items()},
            "kernel_us": {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in km.items()}}


def main():
    order = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    scenes = sys.argv[3:] or ["d2", "r1mix", "tiger", "mmark"]
    for key in scenes:
        wl = bench.Workload(key, 0)
        ring = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
        torch.cuda.synchronize()
        ctxs = []
        for i, letter in enumerate(order):
            e = make_engine(letter, wl.caps)
            e.upload_scene(wl.packed, wl.layout)
            ctxs.append(("%s%d" % (letter, i), e))
        rows = {}
        for rep in range(reps):
            for name, e in ctxs:
                r = measure(e, wl, ring)
                r.update({"scene": key, "context": name, "rep": rep})
                print(json.dumps(r), flush=True)
                rows.setdefault(name, []).append(r)
        imgs = []
        for name, e in ctxs:
            e.render_resident(wl.width, wl.height, WHITE, wl.aa, out=ring[0])
            assert e.sync() == 0
            imgs.append(ring[0].clone())
        same = all(bool(torch.equal(imgs[0], im)) for im in imgs[1:])
        print(json.dumps({"scene": key, "images_equal": same}), flush=True)
        for name, rs in rows.items():
            k = {n: statistics.median([r["kernel_us"][n] for r in rs]) for n in rs[0]["kernel_us"]}
            st = {n: statistics.median([r["stage_us"][n] for r in rs]) for n in rs[0]["stage_us"] if n not in ("flatten", "coarse")}
            sys.stderr.write("%-6s %-4s %6.0f frames/s %6.1f us | %s | %s%s\n" % (
                key, name, statistics.median([f for r in rs for f in r["fps_4_in_flight"]]), statistics.median([r["latency_us"] for r in rs]),
                " ".join("%s %.1f" % (n[2:], x) for n, x in k.items()), " ".join("%s %.1f" % (n[:9], x) for n, x in st.items() if x >= 5.0),
                "" if same else "  IMAGES DIFFER"))
        del ctxs


if __name__ == "__main__":
    main()
