"""Where a stroke workgroup's round goes (measurement build: PROF_FLAGS=-DVELLO_STROKE_TIMELINE scripts/build_prof.sh): per round of 256
stroked lines thread 0's wall-clock stamps -- the line itself (dependent loads list -> tag -> monoid -> points, the arithmetic, the emits
into LDS), the wait for the workgroup's slowest thread, the two reservations + the boxes, the copies out, the last barrier.
    python scripts/stroke_timeline.py [d2] [mmark]          (one frame at a time: k_flatten_main's stroke workgroups; IN_FLIGHT=4: k_flatten_strokes)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("VELLO_PROF_LIB", os.path.join(ROOT, "ab_tmp", "libvello_hip_PROF.so"))
import vello_amd._lib as L
L._use_library(LIB)
import torch
import bench
from vello_amd.renderer import Engine


def report(key):
    wl = bench.Workload(key, 0)
    eng = Engine(0, 1 << int(wl.aa), wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    nif = int(os.environ.get("IN_FLIGHT", "1"))
    eng.set_frames_in_flight(nif)
    out = [torch.zeros((wl.height, wl.width, 4), dtype=torch.uint8, device="cuda") for _ in range(nif)]
    for i in range(4 * nif):
        eng.render_resident(wl.width, wl.height, bench.BASE_COLOR, wl.aa, out=out[i % nif])
    eng.sync()
    lib = ctypes.CDLL(LIB)
    buf = (ctypes.c_uint32 * (8 * 16384))()
    n = ctypes.c_uint32()
    assert lib.vello_stroke_timeline_read(buf, ctypes.byref(n)) == 0  # clears
    eng.render_resident(wl.width, wl.height, bench.BASE_COLOR, wl.aa, out=out[0])
    eng.sync()
    assert lib.vello_stroke_timeline_read(buf, ctypes.byref(n)) == 0
    r = np.array(buf, dtype=np.int64).reshape(-1, 8)[:min(n.value, 16384)]
    if len(r) == 0:
        print(key, "no stroke rounds")
        return
    t = r[:, :7]
    d = lambda a, b: ((t[:, b] - t[:, a]) & 0xffffffff) / 100.0
    span = ((t[:, 6].max() - t[:, 0].min()) & 0xffffffff) / 100.0
    print(f"{key}: {len(r)} rounds in one frame ({nif} in flight configured), launch span {span:.1f} us; lines staged per round mean {(r[:, 7] & 0xffff).mean():.0f}, arcs set aside {(r[:, 7] >> 16).mean():.1f}")
    for name, a, b in (("the line (loads, arithmetic, emits)", 0, 1), ("barrier 1 (the slowest thread)", 1, 2), ("reservations + boxes", 2, 3), ("barrier 2", 3, 4),
                       ("copies out (arcs, lines)", 4, 5), ("barrier 3", 5, 6), ("whole round", 0, 6)):
        x = d(a, b)
        print(f"  {name:40s} mean {x.mean():6.2f} us  p50 {np.median(x):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f}")
    del eng


if __name__ == "__main__":
    for k in sys.argv[1:] or ["d2"]:
        report(k)
