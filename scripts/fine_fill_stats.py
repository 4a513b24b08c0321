#!/usr/bin/env python3
"""What k_fine's fills look like (CPU only, from the oracle's PTCL + segments): per FILL command its segments and pixel-crossing
records (fine.wgsl:186-215: span(x) + span(y) - 1 per segment), per tile the fills in list order.  Prices the candidate
restructurings of k_fine's fill loop: how many passes of 64 record lanes a tile needs when consecutive fills
share a pass (compact counter sets), and how many fills have coverage confined to their record pixels.

    python scripts/fine_fill_stats.py d2|mmark|r1mix|tiger
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import workloads
from oracle.oracle import Oracle



def scene(which):
    if which == "d2":
        return workloads.paris_like_scene_d2().resolve(), 1600, 1600, 2, 8
    if which == "r1mix":
        return workloads.paris_like_scene().resolve(), 1600, 1600, 2, 1
    if which == "mmark":
        return workloads.mmark_scene().resolve(), 2048, 2048, 2, 8
    raise SystemExit("which?")


def span(a, b):
    return np.maximum(np.ceil(np.maximum(a, b)) - np.floor(np.minimum(a, b)), 1.0).astype(np.int64)


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
                # k_coarse's occlusion culling (scenes without clips): an opaque full-tile cover drops everything under it
                while fills_tile and fills_tile[-1] == t:
                    fills_tile.pop(); fills_n.pop(); fills_seg.pop()
            ix += int(_CMD_SIZE[tag])
    ft, fs, fn = np.array(fills_tile), np.array(fills_seg), np.array(fills_n)
    total = int(fn.sum())
    fid = np.repeat(np.arange(fn.size), fn)
    within = np.arange(total) - np.repeat(np.cumsum(fn) - fn, fn)
    s = seg[np.repeat(fs, fn) + within]
    p0x, p0y, p1x, p1y = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    cnt = span(p0x, p1x) + span(p0y, p1y) - 1
    cnt[(p0y == p1y) & (p0y == np.floor(p0y))] = 0
    rec = np.bincount(fid, weights=cnt, minlength=fn.size).astype(np.int64)
    print(f"{which}: tiles {n_tiles}, fills {fn.size} ({fn.size / n_tiles:.1f} a tile), segments {total} ({total / fn.size:.2f} a fill), "
          f"records {rec.sum()} ({rec.mean():.1f} a fill)")
    hist = np.bincount(np.minimum(rec, 200) // 8)
    print("records per fill, histogram by 8:", " ".join(f"{8*i}:{100*v/fn.size:.1f}%" for i, v in enumerate(hist) if v))
    print(f"fills with <= 16 / 32 / 48 / 64 records: {100*(rec<=16).mean():.1f} / {100*(rec<=32).mean():.1f} / {100*(rec<=48).mean():.1f} / {100*(rec<=64).mean():.1f} %")
    print(f"fills with <= 4 / 7 / 8 / 16 segments: {100*(fn<=4).mean():.1f} / {100*(fn<=7).mean():.1f} / {100*(fn<=8).mean():.1f} / {100*(fn<=16).mean():.1f} %")
    # passes: today one per fill (ceil(rec / 64), at least 1); greedy packing of consecutive fills of a tile into passes of <= CAP records
    for cap, maxf in ((64, 2), (64, 4), (64, 8), (128, 4), (128, 8)):
        passes = 0
        cur, nf, last_t = 0, 0, -1
        for t, r in zip(ft, rec):
            r = int(r)
            if t != last_t or nf == maxf or cur + r > cap or r > cap:
                passes += 1 if (t == last_t or last_t == -1 or True) else 0
                cur, nf = 0, 0
                if r > cap:
                    passes += (r + cap - 1) // cap - 1
            cur += r; nf += 1; last_t = t
        print(f"  passes of <= {cap} records, <= {maxf} consecutive fills each: {passes} ({passes / fn.size:.3f} a fill; today {np.maximum((rec + 63) // 64, 1).sum() / fn.size:.3f})")


if __name__ == "__main__":
    main()
