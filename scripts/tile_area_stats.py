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

# Round 6: what a row flag "this row holds a backdrop bump" -- set by k_path_count's flush, read by k_backdrop --
# would let the prefix pass skip, and what a sparse clean-up of the pool (instead of tile_alloc's dense zero fill) would still write
bump_rows = 0
for p in range(n_paths):
    if w[p] == 0 or h[p] == 0:
        continue
    bd = tiles[base[p]:base[p] + w[p] * h[p], 0].reshape(h[p], w[p])
    bump_rows += int((bd != 0).any(axis=1).sum())
print(f"  (path, tile-row) pairs with a non-zero backdrop bump after path_count (the rows k_backdrop has work in): {bump_rows} ({100 * bump_rows / max(rows, 1):.1f} %)")
o.run("backdrop", "backdrop")
tiles2 = o.buffer("tiles", np.int32)[:b["tile"] * 2].reshape(-1, 2)
nz = (tiles2[:, 0] != 0) | (tiles2[:, 1] != 0)
pad2 = np.zeros(lines * 16, dtype=bool)
pad2[:b["tile"]] = nz
l64 = int(pad2.reshape(-1, 16)[: lines // 1].any(axis=1).sum())
l128 = int(pad2[: (lines // 2) * 32].reshape(-1, 32).any(axis=1).sum())
print(f"  after backdrop: non-zero tiles {int(nz.sum())} ({100 * nz.mean():.1f} %); 128-byte lines (16 tiles) holding one: {l64} ({100 * l64 / lines:.1f} %); "
      f"256-byte pieces: {l128} ({100 * l128 / max(lines // 2, 1):.1f} %) -- what a sparse clean-up would still have to write")
