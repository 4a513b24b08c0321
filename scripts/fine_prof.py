"""Per-phase cycle counts of k_fine (measurement build: scripts/build_prof.sh -> ab_tmp/libvello_hip_PROF.so).
Renders a bench workload one frame at a time and prints where the tiles' waves spent their time: over all tiles (what the chip
is busy with) and over the longest tiles (what the launch waits for).   python scripts/fine_prof.py [d2] [r1mix]
The timers cost something themselves (s_memtime waits for the wave's outstanding LDS / scalar loads at every mark, and the 16
accumulators spill): shares are indicative, the frame is ~15 % slower than the product build's."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vello_amd._lib as L
L._use_library(os.environ.get("VELLO_PROF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab_tmp", "libvello_hip_PROF.so")))
import bench
from vello_amd.renderer import Engine

NAMES = ["interpreter", "batch: scan", "batch: segments", "batch: items", "fill: apply", "fill: prefix", "fill: sparse eval",
         "fill: restore", "fill: even-odd", "fill: unbatched", "blend", "rare: begin_clip", "rare: end_clip (blend)",
         "rare: gradients", "rare: image / blur"]
NP = len(NAMES)  # phases; behind them: fills, batches, crossing records, command words, rare commands, fills through ms_fill_simple
SLOTS = NP + 6


def report(key, width=None, height=None, aa=None):
    wl = bench.Workload(key, 0)
    width, height, aa = width or wl.width, height or wl.height, int(wl.aa) if aa is None else aa
    eng = Engine(0, 1 << aa, wl.caps)
    eng.upload_scene(wl.packed, wl.layout)
    report_engine(key, eng, width, height, aa)
    del eng


def report_engine(key, eng, width, height, aa=2):
    """The phase table of the scene `eng` holds (uploaded by the caller: any scene, brushes included)."""
    for _ in range(3):
        eng.render_resident(width, height, bench.BASE_COLOR, aa)
        eng.sync()
    n_tiles = ((width + 15) // 16) * ((height + 15) // 16)
    cap = eng.capacities()["blend_spill"]
    raw = eng.read_buffer("blend_spill", np.uint32)[cap - n_tiles * SLOTS:cap].reshape(n_tiles, SLOTS).astype(np.float64)
    cyc, fills, batches, items, words, rare, simple = raw[:, :NP], raw[:, NP], raw[:, NP + 1], raw[:, NP + 2], raw[:, NP + 3], raw[:, NP + 4], raw[:, NP + 5]
    tot = cyc.sum(axis=1)
    order = np.argsort(tot)
    top = order[-max(1, n_tiles // 256):]  # the slowest 0.4 % of the tiles
    print(f"{key}: {n_tiles} tiles, {fills.sum():.0f} fills in {batches.sum():.0f} batches, {items.sum():.0f} crossing records, "
          f"{words.sum():.0f} command words, {rare.sum():.0f} rare commands; {100 * simple.sum() / max(fills.sum(), 1):.0f} % of the fills through ms_fill_simple")
    print(f"  cycles per tile: mean {tot.mean():.0f}, max {tot.max():.0f} (= {tot.max() / 2400:.0f} us at 2.4 GHz); per fill: "
          f"{tot.sum() / max(fills.sum(), 1):.0f}; slowest tiles: {fills[top].mean():.0f} fills, {words[top].mean():.0f} words, "
          f"{tot[top].sum() / max(fills[top].sum(), 1):.0f} cycles per fill")
    print(f"  {'phase':20s} {'all tiles':>10s} {'slowest':>10s} {'cycles/fill':>12s} {'slowest':>10s}")
    for i, n in enumerate(NAMES):
        print(f"  {n:20s} {100 * cyc[:, i].sum() / tot.sum():9.1f}% {100 * cyc[top, i].sum() / tot[top].sum():9.1f}% "
              f"{cyc[:, i].sum() / max(fills.sum(), 1):12.0f} {cyc[top, i].sum() / max(fills[top].sum(), 1):10.0f}")
    print(f"  per batch: {fills.sum() / max(batches.sum(), 1):.1f} fills, {items.sum() / max(batches.sum(), 1):.0f} records; "
          f"batch build {cyc[:, 1:4].sum() / max(batches.sum(), 1):.0f} cycles")
    if rare.sum():
        print(f"  per rare command: {cyc[:, 11:15].sum() / rare.sum():.0f} cycles ({rare.sum() / n_tiles:.1f} a tile)")


if __name__ == "__main__":
    for k in sys.argv[1:] or ["d2", "r1mix"]:
        report(k)
