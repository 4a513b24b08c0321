#!/usr/bin/env python3
"""How many crossing records of a fill share a pixel (CPU only, the oracle's PTCL + segments, k_coarse's culling replayed as in
fine_fill_stats.py): prices a k_fine fill loop that evaluates a pixel's coverage from its records in registers (one or two
records a pixel) and keeps the LDS sample counters for the pixels with more.

    python scripts/fine_record_stats.py d2|mmark|r1mix
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scripts.fine_fill_stats import scene, span
from oracle.oracle import Oracle


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "d2"
    (packed, layout), w, h, aa, scale = scene(which)
    o = Oracle(capacity_scale=scale)
    o.set_threads(8)
    o.set_scene(packed, layout, w, h, 0xFFFFFFFF, aa)
    o.render()
    ptcl = o.buffer("ptcl", np.uint32)
    seg = o.buffer("segments", np.float32).reshape(-1, 6)
    n_tiles = ((w + 15) // 16) * ((h + 15) // 16)
    from tests.parity import CMD_JUMP as PJ, _CMD_SIZE
    fills_tile, fills_seg, fills_n = [], [], []
    for t in range(n_tiles):
        ix = t * 64 + 1
        while True:
            tag = int(ptcl[ix])
            if tag == 0:
                break
            if tag == PJ:
                ix = int(ptcl[ix + 1])
                continue
            if tag == 1:
                fills_tile.append(t); fills_n.append(int(ptcl[ix + 1]) >> 1); fills_seg.append(int(ptcl[ix + 2]))
            if tag == 3 and int(ptcl[ix + 1]) == 5 and (int(ptcl[ix + 2]) >> 24) == 0xff:
                while fills_tile and fills_tile[-1] == t:
                    fills_tile.pop(); fills_n.pop(); fills_seg.pop()
            ix += int(_CMD_SIZE[tag])
    fs, fn = np.array(fills_seg), np.array(fills_n)
    total = int(fn.sum())
    fid = np.repeat(np.arange(fn.size), fn)
    within = np.arange(total) - np.repeat(np.cumsum(fn) - fn, fn)
    s = seg[np.repeat(fs, fn) + within].astype(np.float32)
    p0x, p0y, p1x, p1y = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    f32 = np.float32
    down = p1y >= p0y
    x0 = np.where(down, p0x, p1x); y0 = np.where(down, p0y, p1y)
    x1 = np.where(down, p1x, p0x); y1 = np.where(down, p1y, p0y)
    dx = np.abs(x1 - x0); dy = y1 - y0
    with np.errstate(all="ignore"):
        idxdy = f32(1.0) / (dx + dy)
        a = dx * idxdy
        pos = x1 >= x0
        sgn = np.where(pos, f32(1.0), f32(-1.0))
        xt0 = np.floor(x0 * sgn)
        c = x0 * sgn - xt0
        y0i = np.floor(y0)
        b = np.minimum((dy * c + dx * (y0i + 1 - y0)) * idxdy, f32(1.0 - 2.0 ** -24))
    count_x = span(x0, x1) - 1
    cnt = count_x + span(y0, y1)
    cnt[(p0y == p1y) & (p0y == np.floor(p0y))] = 0
    x0i = (xt0 * sgn + f32(0.5) * (sgn - 1)).astype(np.int64)
    # one row per record
    rfid = np.repeat(fid, cnt)
    rseg = np.repeat(np.arange(total), cnt)
    sub = (np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)).astype(np.float32)
    z = np.floor(a[rseg] * sub + b[rseg])
    x = x0i[rseg] + (np.where(pos[rseg], z, -z)).astype(np.int64)
    y = y0i[rseg].astype(np.int64) + sub.astype(np.int64) - z.astype(np.int64)
    ok = (x >= 0) & (x < 16) & (y >= 0) & (y < 16)
    key = rfid[ok] * 256 + y[ok] * 16 + x[ok]
    uk, mult = np.unique(key, return_counts=True)
    n_rec = int(ok.sum())
    print(f"{which}: fills {fn.size}, records in the tile {n_rec} ({n_rec / fn.size:.1f} a fill), distinct (fill, pixel) {uk.size} ({uk.size / fn.size:.1f} a fill)")
    h_ = np.bincount(np.minimum(mult, 9))
    print("pixels by records on them: " + " ".join(f"{i}{'+' if i == 9 else ''}:{100 * v / uk.size:.1f}%" for i, v in enumerate(h_) if v))
    print("records on pixels with 1 / 2 / >= 3 records: " + " / ".join(f"{100 * (mult[sel] .sum()) / n_rec:.1f}" for sel in (mult == 1, mult == 2, mult >= 3)) + " %")
    fmax = np.zeros(fn.size, dtype=np.int64)
    np.maximum.at(fmax, uk // 256, mult)
    print("fills by the largest number of records on one pixel: " + " ".join(f"{i}:{100 * (fmax == i).mean():.1f}%" for i in range(0, 6)) + f" >=6:{100 * (fmax >= 6).mean():.1f}%")
    # greedy pairs of consecutive fills of a tile's list: both <= 32 records, no pixel of one touched by the other -- the pairs that
    # could share ONE pass over the pixel-indexed sample counters (a half-wave of record lanes each)
    ft = np.array(fills_tile)
    nrec = np.bincount(rfid[ok], minlength=fn.size)
    pix_sets = np.split(uk % 256, np.cumsum(np.bincount(uk // 256, minlength=fn.size))[:-1])
    paired = 0; candidates = 0; collide = 0
    i = 0
    while i + 1 < fn.size:
        if ft[i] == ft[i + 1] and 0 < nrec[i] <= 32 and 0 < nrec[i + 1] <= 32:
            candidates += 1
            if np.intersect1d(pix_sets[i], pix_sets[i + 1], assume_unique=True).size == 0:
                paired += 2; i += 2; continue
            collide += 1
        i += 1
    print(f"greedy pairs of consecutive fills (both <= 32 records): {100 * paired / fn.size:.1f} % of the fills end up in a pair with disjoint pixels; "
          f"{collide} of {candidates} candidate pairs share a pixel")
    npx = np.bincount(uk // 256, minlength=fn.size)
    print(f"touched pixels per fill: mean {npx.mean():.1f}, <= 16 / 32 / 64: {100 * (npx <= 16).mean():.1f} / {100 * (npx <= 32).mean():.1f} / {100 * (npx <= 64).mean():.1f} %")


if __name__ == "__main__":
    main()
