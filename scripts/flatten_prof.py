"""Where the heavy flatten kernels spend their wave cycles (a phase profile BEFORE building anything).
Measurement build: PROF_FLAGS="-DVELLO_FLATTEN_PROF" bash scripts/build_prof.sh -> ab_tmp/libvello_hip_PROF.so.
   python scripts/flatten_prof.py [mmark] [tiger] [d2]
One frame at a time.  The timers are per WAVE (flatten.hip: FlProfLds): cycles = wave cycles between marks, passages = how
often the wave executed that code (the union of its lanes' loops); lane-level counts beside them give the lane use."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("VELLO_PROF_LIB", os.path.join(ROOT, "ab_tmp", "libvello_hip_PROF.so"))
import vello_amd._lib as L
L._use_library(LIB)
import bench
from vello_amd.renderer import Engine, STAGES

PHASES = ["tag decode + loads", "subdivision turn", "accepted piece setup", "euler line", "straight shortcut", "join / cap",
          "arc setup", "arc line", "bbox + flush", "other"]
COUNTS = ["entries", "subdivision turns", "pieces (x sides)", "euler lines", "arcs", "arc lines"]


def report(key):
    wl = bench.Workload(key, 0)
    eng = Engine(0, 1 << int(wl.aa), wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    for _ in range(3):
        eng.render_resident(wl.width, wl.height, bench.BASE_COLOR, wl.aa)
        eng.sync()
    lib = ctypes.CDLL(LIB)
    n = 2 * len(PHASES) + len(COUNTS) + 2
    buf = (ctypes.c_ulonglong * n)()
    assert lib.vello_flatten_prof_read(buf) == 0  # clears
    eng.set_profiling(STAGES)
    eng.stage_ms(); eng.kernel_ms()
    reps = 5
    for _ in range(reps):
        eng.render_resident(wl.width, wl.height, bench.BASE_COLOR, wl.aa)
        eng.sync()
    st, km = eng.stage_ms(), eng.kernel_ms()
    assert lib.vello_flatten_prof_read(buf) == 0
    v = np.array(list(buf), dtype=np.float64) / reps
    P = len(PHASES)
    cyc, pas, cnt, waves, wave_cyc = v[:P], v[P:2 * P], v[2 * P:2 * P + len(COUNTS)], v[-2], v[-1]
    bump = eng.bump()
    print(f"{key}: {bump['lines']} lines; flatten stage {st['flatten'][0] / max(st['flatten'][1], 1) * 1e3:.1f} us; "
          + ", ".join(f"{k} {ms / max(c, 1) * 1e3:.1f} us" for k, (ms, c) in km.items() if k.startswith("k_flatten")))
    print(f"  heavy waves {waves:.0f}, mean wave {wave_cyc / max(waves, 1):.0f} cycles ({wave_cyc / max(waves, 1) / 2400:.1f} us at 2.4 GHz); "
          f"marked {cyc.sum() / max(wave_cyc, 1) * 100:.0f} % of the wave cycles")
    print(f"  {'phase':22s} {'share':>7s} {'cycles/wave':>12s} {'passages/wave':>14s} {'cycles/passage':>15s}")
    for i, nme in enumerate(PHASES):
        print(f"  {nme:22s} {100 * cyc[i] / max(cyc.sum(), 1):6.1f}% {cyc[i] / max(waves, 1):12.0f} {pas[i] / max(waves, 1):14.1f} {cyc[i] / max(pas[i], 1):15.0f}")
    print("  lane-level counts: " + ", ".join(f"{n} {c:.0f}" for n, c in zip(COUNTS, cnt)))
    pairs = [("subdivision turns", 1, 1), ("pieces (x sides)", 2, 2), ("euler lines", 3, 3), ("arcs", 6, 4), ("arc lines", 7, 5)]
    print("  lane use (lane-level count / 64 x passages): " + ", ".join(f"{n} {100 * cnt[ci] / max(64 * pas[pi], 1):.0f} %" for n, pi, ci in pairs))
    del eng


if __name__ == "__main__":
    for k in sys.argv[1:] or ["mmark", "tiger", "d2"]:
        report(k)
