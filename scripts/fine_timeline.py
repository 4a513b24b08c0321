"""Occupancy of the chip over one k_fine launch (measurement build: PROF_FLAGS=-DVELLO_FINE_TIMELINE scripts/build_prof.sh ->
ab_tmp/libvello_hip_PROF.so).  Every wave logs its start / end (wall clock, 100 MHz), its HW_ID and what it was: a tile
(0), a slice (1), a slice followed by its tile's compositing pass (2).   python scripts/fine_timeline.py [d2] [r1mix]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vello_amd._lib as L
L._use_library(os.environ.get("VELLO_PROF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_tmp", "libvello_hip_PROF.so")))
import bench
from vello_amd.renderer import Engine

TL_MAX = 40000


def report(key, slices=True):
    wl = bench.Workload(key, 0)
    eng = Engine(0, 4, wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    for _ in range(3):
        eng.render_resident(bench.WIDTH, bench.HEIGHT, bench.BASE_COLOR, 2)
        eng.sync()
    cap = eng.capacities()["blend_spill"]
    zero = np.zeros(6 * TL_MAX + 1, dtype=np.uint32)
    eng.write_buffer("blend_spill", zero, (cap - 1 - 6 * TL_MAX) * 4)
    eng.render_resident(bench.WIDTH, bench.HEIGHT, bench.BASE_COLOR, 2)
    eng.sync()
    raw = eng.read_buffer("blend_spill", np.uint32)[cap - 1 - 6 * TL_MAX:cap]
    n = int(raw[-1])
    log = raw[:6 * min(n, TL_MAX)].reshape(-1, 6).astype(np.int64)
    t0, t1, hw, kind, tile, xcc = log[:, 0], log[:, 1], log[:, 2], log[:, 3] & 0xff, log[:, 4], log[:, 5] & 0xf
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0  # us
    dur = t1 - t0
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)  # cu_id, sh_id, se_id, xcc
    simd = (hw >> 4) & 3
    print(f"{key}: {n} waves logged; launch span {t1.max():.1f} us; wave time: sum {dur.sum():.0f} us, mean {dur.mean():.2f}, max {dur.max():.1f}")
    for k, name in ((0, "tile"), (1, "slice"), (2, "slice+composite")):
        m = kind == k
        if m.any():
            print(f"  {name:16s} {m.sum():6d} waves, wave time sum {dur[m].sum():9.0f} us, mean {dur[m].mean():7.2f}, max {dur[m].max():7.1f}, last end {t1[m].max():.1f}")
    print(f"  distinct CUs {len(np.unique(cu))}, distinct (CU, SIMD) {len(np.unique(cu * 4 + simd))}")
    # waves resident over time
    step = 5.0
    edges = np.arange(0.0, t1.max() + step, step)
    print("  time us : resident waves (mean over the bin) | waves started in the bin")
    for a in edges:
        b = a + step
        overlap = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0.0, None).sum() / step
        started = int(((t0 >= a) & (t0 < b)).sum())
        print(f"  {a:6.0f}  : {overlap:7.0f} | {started}")
    # the waves that end last
    order = np.argsort(t1)[-8:]
    for i in order:
        print(f"  late wave: kind {kind[i]} tile {tile[i]} start {t0[i]:.1f} end {t1[i]:.1f} ({dur[i]:.1f} us)")
    del eng


if __name__ == "__main__":
    for k in sys.argv[1:] or ["d2"]:
        report(k)
