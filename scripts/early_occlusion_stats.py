#!/usr/bin/env python3
"""Prices occlusion known BEFORE path_count: CPU only, from the oracle's buffers of the road-map scene d2.

k_coarse drops, per batch of 256 queued draw objects, what lies under a tile's opaque full cover; everything upstream of it
(path_count's tile atomics and SegmentCount records, path_tiling's whole pass) has by then been paid for every crossing.  This
script asks what a two-phase front would save: phase A counts and prefix-sums only the OCCLUDER CANDIDATES (fills with an
opaque solid colour outside clips), a pass over their tile rectangles leaves `occ[screen tile]` = the draw index of the tile's
last full cover, and phase B runs the other paths with a test `draw_ix < occ[tile]` per crossing (a dead crossing keeps its
backdrop bump -- the row's prefix sum runs on to live tiles -- and loses its segment count and its SegmentCount record).

Per FILL command of the oracle's PTCL (no culling there: the reference's lists): the pool tile from its segment slice, the path
from the tile, the path's kind from the generator (paris_like_scene_d2: kinds < 0.70 are strokes).

    python scripts/early_occlusion_stats.py            # d2, seed 0x5EED0001
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import workloads  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.parity import CMD_JUMP, _CMD_SIZE  # noqa: E402

CMD_END, CMD_FILL, CMD_SOLID, CMD_COLOR = 0, 1, 3, 5


def main():
    seed, n_paths = 0x5EED0001, 30000
    packed, layout = workloads.paris_like_scene_d2(seed).resolve()
    w = h = 1600
    o = Oracle(capacity_scale=8)
    o.set_threads(8)
    o.set_scene(packed, layout, w, h, 0xFFFFFFFF, 2)
    o.render()
    bump = o.bump()
    ptcl = o.buffer("ptcl", np.uint32)
    tiles = o.buffer("tiles", np.uint32).reshape(-1, 2)[: bump["tile"]]
    paths = o.buffer("paths", np.uint32).reshape(-1, 8)[:n_paths]
    lines = o.buffer("lines", np.uint32).reshape(-1, 6)[: bump["lines"]]
    is_stroke = np.random.Generator(np.random.PCG64(seed)).random(n_paths) < 0.70  # (the generator's first draw)

    # pool tile of a segment slice: coarse.wgsl:392 leaves ~seg_start in the tile
    seg_start = (~tiles[:, 1]).astype(np.int64)
    live = tiles[:, 1].astype(np.int32) < 0
    start_to_tile = dict(zip(seg_start[live].tolist(), np.nonzero(live)[0].tolist()))
    path_first = paths[:, 4].astype(np.int64)
    order = np.argsort(path_first, kind="stable")

    def path_of_tile(t):
        return int(order[np.searchsorted(path_first[order], t, side="right") - 1])

    n_tx, n_ty = (w + 15) // 16, (h + 15) // 16
    tot = np.zeros(2, np.int64)          # crossings by kind (0 fill, 1 stroke)
    dead = np.zeros(2, np.int64)         # ... under the tile's LAST full cover (complete culling)
    fills_all = fills_live = 0
    pairs_dead = 0
    for t in range(n_tx * n_ty):
        ix = t * 64 + 1
        cmds = []   # (n_segs, kind) per FILL, in list order; occluder positions
        last_occ = -1
        pending_solid = False
        while True:
            tag = int(ptcl[ix])
            if tag == CMD_END:
                break
            if tag == CMD_JUMP:
                ix = int(ptcl[ix + 1])
                continue
            if tag == CMD_FILL:
                n = int(ptcl[ix + 1]) >> 1
                p = path_of_tile(start_to_tile[int(ptcl[ix + 2])])
                cmds.append((n, 1 if is_stroke[p] else 0))
                pending_solid = False
            elif tag == CMD_SOLID:
                pending_solid = True
            elif tag == CMD_COLOR:
                if pending_solid and (int(ptcl[ix + 1]) >> 24) == 0xFF:
                    last_occ = len(cmds)   # everything in cmds[:last_occ] lies under it
                pending_solid = False
            else:
                pending_solid = False
            ix += int(_CMD_SIZE[tag])
        for i, (n, k) in enumerate(cmds):
            tot[k] += n
            if i < last_occ:
                dead[k] += n
                pairs_dead += 1
        fills_all += len(cmds)
        fills_live += len(cmds) - max(last_occ, 0)
    C = int(tot.sum())
    print(f"d2 seed {seed:#x}: crossings in the lists {C} (bump.seg_counts {bump['seg_counts']}), FILL commands {fills_all}")
    print(f"under a tile's last opaque full cover (complete culling): {int(dead.sum())} crossings = {100 * dead.sum() / C:.1f} %, "
          f"{pairs_dead} (tile, path) pairs; FILLs that reach fine {fills_live}, their segments {C - int(dead.sum())}")
    print(f"  strokes: {int(tot[1])} crossings ({100 * tot[1] / C:.1f} % of all), dead {int(dead[1])} = {100 * dead[1] / max(tot[1], 1):.1f} % of theirs")
    print(f"  fills  : {int(tot[0])} crossings ({100 * tot[0] / C:.1f} % of all), dead {int(dead[0])} = {100 * dead[0] / max(tot[0], 1):.1f} % of theirs")
    lp = lines[:, 0]
    lp = lp[lp < n_paths]
    cand_lines = int((~is_stroke[lp]).sum())
    area = (paths[:, 2] - paths[:, 0]).astype(np.int64) * (paths[:, 3] - paths[:, 1]).astype(np.int64)
    print(f"phase A (the fills): {cand_lines} of {lines.shape[0]} lines = {100 * cand_lines / lines.shape[0]:.1f} %, "
          f"{int(area[~is_stroke].sum())} of {int(area.sum())} pool tiles = {100 * area[~is_stroke].sum() / area.sum():.1f} %")
    saved = int(dead[1])
    print(f"phase B would drop {saved} SegmentCount records ({saved * 8 / 1e6:.1f} MB written + read), the same number of path_tiling "
          f"threads ({100 * saved / C:.1f} % of that pass) and of tile-count atomics' addends; the fills' own dead crossings ({int(dead[0])}) "
          f"stay with k_coarse as today")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def lines_under_covers():
    """What occlusion known before FLATTEN could leave out: the lines of d2's soup ALL of whose tiles (the tiles of the line's
    bounding box: what a test per line could afford) lie under a later fill's opaque full cover.  occ[] as k_occ_build would make it: per
    fill, the tiles of its rectangle without segments and with a backdrop that is not clear."""
    seed, n_paths = 0x5EED0001, 30000
    packed, layout = workloads.paris_like_scene_d2(seed).resolve()
    w = h = 1600
    o = Oracle(capacity_scale=8)
    o.set_threads(8)
    o.set_scene(packed, layout, w, h, 0xFFFFFFFF, 2)
    o.render()
    bump = o.bump()
    tiles = o.buffer("tiles", np.int32).reshape(-1, 2)[: bump["tile"]]
    paths = o.buffer("paths", np.uint32).reshape(-1, 8)[:n_paths]
    lines_u = o.buffer("lines", np.uint32).reshape(-1, 6)[: bump["lines"]]
    lines_f = o.buffer("lines", np.float32).reshape(-1, 6)[: bump["lines"]]
    is_stroke = np.random.Generator(np.random.PCG64(seed)).random(n_paths) < 0.70
    n_tx, n_ty = (w + 15) // 16, (h + 15) // 16
    occ = np.zeros((n_ty, n_tx), np.int64)
    for i in np.nonzero(~is_stroke)[0]:   # (every colour of the scene is opaque; the fills are non-zero)
        x0, y0, x1, y1, first = (int(v) for v in paths[i, :5])
        if x1 <= x0 or y1 <= y0:
            continue
        t = tiles[first: first + (x1 - x0) * (y1 - y0)].reshape(y1 - y0, x1 - x0, 2)
        cover = (t[:, :, 1] == 0) & (t[:, :, 0] != 0)
        sub = occ[y0:y1, x0:x1]
        sub[cover] = np.maximum(sub[cover], i + 1)
    print(f"tiles with an opaque full cover: {int((occ > 0).sum())} of {n_tx * n_ty}")
    p = lines_u[:, 0].astype(np.int64)
    ok = p < n_paths
    xs0 = np.floor(np.minimum(lines_f[:, 2], lines_f[:, 4]) / 16).astype(np.int64)
    xs1 = np.floor(np.maximum(lines_f[:, 2], lines_f[:, 4]) / 16).astype(np.int64)
    ys0 = np.floor(np.minimum(lines_f[:, 3], lines_f[:, 5]) / 16).astype(np.int64)
    ys1 = np.floor(np.maximum(lines_f[:, 3], lines_f[:, 5]) / 16).astype(np.int64)
    inside = ok & (xs0 >= 0) & (ys0 >= 0) & (xs1 < n_tx) & (ys1 < n_ty)
    # min of occ over the line's box of tiles, by a running minimum over the (few) tiles of the box
    dead = np.zeros(lines_u.shape[0], bool)
    idx = np.nonzero(inside)[0]
    span = (xs1[idx] - xs0[idx] + 1) * (ys1[idx] - ys0[idx] + 1)
    small = idx[span <= 16]
    m = np.full(small.size, np.iinfo(np.int64).max)
    for dy in range(4):
        for dx in range(16):
            xx, yy = xs0[small] + dx, ys0[small] + dy
            use = (xx <= xs1[small]) & (yy <= ys1[small])
            v = occ[np.minimum(yy, n_ty - 1), np.minimum(xx, n_tx - 1)]
            m = np.where(use, np.minimum(m, v), m)
    dead[small] = m > p[small] + 1
    off = ok & ~inside & ((xs1 < 0) | (ys1 < 0) | (xs0 >= n_tx) | (ys0 >= n_ty))
    n = lines_u.shape[0]
    st = ok & is_stroke[np.minimum(p, n_paths - 1)]
    print(f"lines {n}: of strokes {int(st.sum())}; every tile of the line's box under a later cover: {int(dead.sum())} = {100 * dead.sum() / n:.1f} % "
          f"(strokes' {int((dead & st).sum())} = {100 * (dead & st).sum() / max(st.sum(), 1):.1f} % of theirs); wholly off the target {int(off.sum())}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "lines":
    lines_under_covers()
