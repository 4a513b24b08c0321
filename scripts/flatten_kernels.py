"""Flatten's kernels one by one (HIP events between the launches, one frame at a time) on d2 / r1mix / mmark / tiger.
   python scripts/flatten_kernels.py [A|<variant letter>] [workloads ...]     (ab_tmp/libvello_hip_<letter>.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "A"
if which != "A":
    import vello_amd._lib as L
    L._use_library(os.path.join(ROOT, "ab_tmp", f"libvello_hip_{which}.so"))
import bench
from vello_amd.renderer import Engine

for key in (sys.argv[2:] or ("d2", "r1mix", "mmark", "tiger")):
    wl = bench.Workload(key, 0)
    eng = Engine(0, 1 << int(wl.aa), wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    for _ in range(5):
        eng.render_resident(wl.width, wl.height, bench.BASE_COLOR, wl.aa); eng.sync()
    eng.set_profiling(["flatten"])
    eng.stage_ms(); eng.kernel_ms()
    for _ in range(30):
        eng.render_resident(wl.width, wl.height, bench.BASE_COLOR, wl.aa); eng.sync()
    st, km = eng.stage_ms(), eng.kernel_ms()
    print(which, key, "flatten %.1f us:" % (st["flatten"][0] / max(st["flatten"][1], 1) * 1e3),
          ", ".join("%s %.1f" % (k, ms / max(c, 1) * 1e3) for k, (ms, c) in km.items() if k.startswith("k_flatten")), flush=True)
    del eng
