"""Where k_path_count's workgroups spend their time (measurement build: PROF_FLAGS=-DVELLO_PC_TIMELINE scripts/build_prof.sh):
per chunk of 1024 lines the wall-clock stamps of its phases.
    python scripts/pc_timeline.py [d2] [r1mix]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vello_amd._lib as L
L._use_library(os.environ.get("VELLO_PROF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_tmp", "libvello_hip_PROF.so")))
import bench
from vello_amd.renderer import Engine


def report_tiger():
    """C2: the tiger at 1024 x 1024 -- a soup of 15 000 lines, k_path_count<1>: chunks of 256 lines, one per workgroup"""
    import vello_amd
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiger_scene.npz"))
    eng = Engine()
    eng.upload_scene(d["packed"], vello_amd.Layout(*[int(v) for v in d["layout"]]))
    for _ in range(3):
        eng.render_resident(1024, 1024, 0xFFFFFFFF, 1)
        eng.sync()
    cap = eng.capacities()["seg_counts"]
    n_chunks = (eng.bump()["lines"] + 255) // 256
    raw = eng.read_buffer("seg_counts", np.uint32)[(cap - 4 * 8192) * 2:cap * 2].reshape(-1, 8).astype(np.int64)[:n_chunks]
    report_chunks("tiger", raw, n_chunks)


def report(key):
    if key == "tiger":
        return report_tiger()
    wl = bench.Workload(key, 0)
    eng = Engine(0, 4, wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    for _ in range(3):
        eng.render_resident(bench.WIDTH, bench.HEIGHT, bench.BASE_COLOR, 2)
        eng.sync()
    cap = eng.capacities()["seg_counts"]
    n_lines = eng.bump()["lines"]
    n_chunks = (n_lines + 1023) // 1024
    raw = eng.read_buffer("seg_counts", np.uint32)[(cap - 4 * 8192) * 2:cap * 2].reshape(-1, 8).astype(np.int64)[:n_chunks]
    report_chunks(key, raw, n_chunks)
    del eng


def report_chunks(key, raw, n_chunks):
    """start / walks + scan done / counted (pass A) / end, flush answered / records written (pass B), table entries
    used, crossings that went to memory directly."""
    t0, t1, t2, t3, t4, t5, n_occ, direct = (raw[:, i] for i in range(8))
    base = t0.min()
    us = lambda t: (t - base) / 100.0
    print(f"{key}: {n_chunks} chunks; launch span {us(t3).max():.1f} us; table entries used per chunk mean {n_occ.mean():.0f} p90 {np.percentile(n_occ, 90):.0f} "
          f"max {n_occ.max()}; crossings sent to memory directly {direct.sum()} ({direct.sum() / max(n_chunks, 1):.1f} per chunk)")
    for name, a, b in (("pass 1 (loads, walks, scan)", t0, t1), ("pass A (count in LDS)", t1, t2), ("flush (atomics answered)", t2, t4),
                       ("pass B (records)", t4, t5), ("clear", t5, t3), ("whole chunk", t0, t3)):
        d = (b - a) / 100.0
        print(f"  {name:28s} mean {d.mean():7.2f} us  p50 {np.median(d):7.2f}  p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f}   sum {d.sum():9.0f}")
    step = 5.0
    for a in np.arange(0.0, us(t3).max() + step, step):
        b = a + step
        ov = lambda x, y: np.clip(np.minimum(us(y), b) - np.maximum(us(x), a), 0, None).sum() / step
        print(f"  {a:6.0f} us: resident chunks {ov(t0, t3):6.0f}  (pass 1 {ov(t0, t1):6.0f}, A {ov(t1, t2):6.0f}, flush {ov(t2, t4):6.0f}, B {ov(t4, t5):6.0f}) "
              f"started {int(((us(t0) >= a) & (us(t0) < b)).sum())}")


if __name__ == "__main__":
    for k in sys.argv[1:] or ["d2"]:
        report(k)
