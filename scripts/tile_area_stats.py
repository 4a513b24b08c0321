"""How much of the path-tile area does path_count touch ("a per-(path, tile-row) touched bit so that k_backdrop
and k_coarse_prep skip untouched rows")?  CPU only: the oracle's stages up to path_count on the bench's d2 scene, then per path the
rows of its tile rectangle, the 16-tile cache lines of the pool, and the tiles that hold a crossing or a backdrop bump.
   python scripts/tile_area_stats.py            -> profiles/r05_tile_area.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle.oracle import Oracle

wl = bench.Workload("d2", 0)
o = Oracle(capacity_scale=8, auto_grow=True)
o.set_threads(os.cpu_count() or 1)
o.set_scene(wl.packed, wl.layout, wl.width, wl.height, bench.BASE_COLOR, int(wl.aa))
o.run("pathtag_scan", "path_count")
b = o.bump()
n_paths = wl.layout.n_paths
paths = o.buffer("paths", np.uint32)[:n_paths * 8].reshape(n_paths, 8)
tiles = o.buffer("tiles", np.int32)[:b["tile"] * 2].reshape(-1, 2)
touched = (tiles[:, 0] != 0) | (tiles[:, 1] != 0)
w = (paths[:, 2] - paths[:, 0]).astype(np.int64)
h = (paths[:, 3] - paths[:, 1]).astype(np.int64)
base = paths[:, 4].astype(np.int64)
rows = rows_touched = 0
for p in range(n_paths):
    if w[p] == 0 or h[p] == 0:
        continue
    t = touched[base[p]:base[p] + w[p] * h[p]].reshape(h[p], w[p])
    rows += h[p]
    rows_touched += int(t.any(axis=1).sum())
lines = (b["tile"] + 15) // 16
pad = np.zeros(lines * 16, dtype=bool)
pad[:b["tile"]] = touched
lines_touched = int(pad.reshape(-1, 16).any(axis=1).sum())
print(f"d2 ({wl.width} x {wl.height}, {n_paths} paths): {b['tile']} path tiles, {b['seg_counts']} crossings")
print(f"  tiles with a crossing or a backdrop bump after path_count: {int(touched.sum())} ({100 * touched.mean():.1f} %)")
print(f"  (path, tile-row) pairs: {rows}, with a touched tile: {rows_touched} ({100 * rows_touched / max(rows, 1):.1f} %) -- mean rectangle "
      f"{w[w > 0].mean():.1f} x {h[h > 0].mean():.1f} tiles")
print(f"  16-tile cache lines of the pool: {lines}, with a touched tile: {lines_touched} ({100 * lines_touched / max(lines, 1):.1f} %)")
