"""What k_path_count asks of the memory system, counted on the engine's own line soup (numpy; run on the GPU box or, slowly,
on the emulator): scattered atomics cost one request per (instruction, distinct cache line) at 2.7e10 / s chip-wide
(scripts/calib/atomic_rate.hip, atomic_scope.hip), so the question is how many such requests a frame needs

  now:        per wave instruction (64 consecutive lines, crossing step s) the run heads' distinct lines, segment counts and
              backdrop bumps (pair-cancelled between adjacent lanes) as two instructions
  aggregated: the crossings of a whole workgroup chunk (1 024 lines) added up per tile first -- one returning add per touched
              tile, issued 16 tiles of a cache line per 16 lanes -- and the backdrop bumps likewise, zero sums dropped

    python scripts/pc_requests.py [d2|r1mix|tiger|mmark] [chunk_lines]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PC_EMU") == "1":
    import vello_amd._lib as L
    L._use_library(os.path.join(ROOT, "tests", "simt_emu", "libvello_emu.so"))
import vello_amd  # noqa: E402
import bench  # noqa: E402

f32 = np.float32


def span(a, b):
    return np.maximum(np.ceil(np.maximum(a, b)) - np.floor(np.minimum(a, b)), f32(1)).astype(np.uint32)


def walks(lines, paths):
    """Vectorised path_count.wgsl:51-170 (f32 throughout).  Returns per line: valid, imin, imax and the walk's parameters."""
    path_ix = lines[:, 0]
    p = lines[:, 2:6].view(np.float32)
    p0x, p0y, p1x, p1y = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    is_down = p1y >= p0y
    x0_, y0_ = np.where(is_down, p0x, p1x), np.where(is_down, p0y, p1y)
    x1_, y1_ = np.where(is_down, p1x, p0x), np.where(is_down, p1y, p0y)
    s0x, s0y, s1x, s1y = x0_ * f32(0.0625), y0_ * f32(0.0625), x1_ * f32(0.0625), y1_ * f32(0.0625)
    count_x = span(s0x, s1x) - 1
    count = count_x + span(s0y, s1y)
    dx = np.abs(s1x - s0x)
    dy = s1y - s0y
    valid = ~((dx + dy == 0) | ((dy == 0) & (np.floor(s0y) == s0y)))
    with np.errstate(all="ignore"):
        idxdy = f32(1) / (dx + dy)
        a = dx * idxdy
        pos = s1x >= s0x
        sign = np.where(pos, f32(1), f32(-1))
        xt0 = np.floor(s0x * sign)
        c = s0x * sign - xt0
        y0 = np.floor(s0y)
        ytop = np.where(s0y == s1y, np.ceil(s0y), y0 + f32(1))
        b = np.minimum((dy * c + dx * (ytop - s0y)) * idxdy, f32(0.99999994))
        err = np.floor(a * (count.astype(np.float32) - f32(1)) + b) - count_x.astype(np.float32)
        a = np.where(err != 0, a - f32(2e-7) * np.sign(err), a).astype(np.float32)
        x0 = xt0 * sign + np.where(pos, f32(0), f32(-1))
    bbox = paths[path_ix][:, 0:4].astype(np.int64)
    tiles_base = paths[path_ix][:, 4].astype(np.int64)
    stride = bbox[:, 2] - bbox[:, 0]
    valid &= stride > 0
    return dict(valid=valid, count=count, a=a, b=b, x0=x0, y0=y0, sign=sign, s0y=s0y, is_down=is_down, bbox=bbox, tiles_base=tiles_base,
                stride=stride)


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "d2"
    chunk_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    wl = bench.Workload(key, 0)
    eng = vello_amd.Engine(capacities=wl.caps)
    eng.set_frames_in_flight(int(os.environ.get("PC_IN_FLIGHT", "1")))
    eng.upload_scene(wl.packed, wl.layout)
    eng.render_resident(wl.width, wl.height, 0xFF000000, wl.aa)
    eng.sync()
    bump = eng.bump()
    n = bump["lines"]
    lines = eng.read_buffer("lines", np.uint32, n * 24).reshape(n, 6)
    paths = eng.read_buffer("paths", np.uint32, wl.layout.n_paths * 32).reshape(-1, 8)
    w = walks(lines, paths)
    # every crossing: (line, i) -> tile index, top-edge backdrop bump target; the clipping to the path's rectangle is done by
    # dropping what falls outside (the kernel computes imin / imax instead)
    cnt = np.where(w["valid"], w["count"], 0).astype(np.int64)
    cnt = np.minimum(cnt, 4096)
    line_of = np.repeat(np.arange(n), cnt)
    first = np.cumsum(cnt) - cnt
    i = (np.arange(line_of.size) - first[line_of]).astype(np.float32)
    a, b = w["a"][line_of], w["b"][line_of]
    z = np.floor(a * i + b)
    zprev = np.floor(a * (i - f32(1)) + b)
    y = (w["y0"][line_of] + i - z).astype(np.int64)
    x = (w["x0"][line_of] + w["sign"][line_of] * z).astype(np.int64)
    bbox = w["bbox"][line_of]
    inside = (y >= bbox[:, 1]) & (y < bbox[:, 3]) & (x < bbox[:, 2])
    xc = np.maximum(x, bbox[:, 0])
    counted = inside & (x >= bbox[:, 0])
    base = w["tiles_base"][line_of] + (y - bbox[:, 1]) * w["stride"][line_of] - bbox[:, 0]
    tile = base + x
    top_edge = np.where(i == 0, w["y0"][line_of] == w["s0y"][line_of], zprev == z)
    bump_ok = inside & top_edge & (x + 1 < bbox[:, 2])
    btile = base + np.maximum(x + 1, bbox[:, 0])
    delta = np.where(w["is_down"][line_of], -1, 1)
    print(f"{key}: {n} lines, crossings counted {int(counted.sum())} (bump.seg_counts {bump['seg_counts']}), backdrop bumps {int(bump_ok.sum())}")
    for line_shift, name in ((4, "128-byte lines"), (3, "64-byte lines")):
        # now: per (chunk, j, wave, step) instruction
        step = i.astype(np.int64)
        wave_of_line = line_of // 64  # 64 consecutive lines = one wave's lanes at one j
        inst = wave_of_line * 4096 + step
        # run heads: a lane whose left neighbour (line - 1, same wave, same step, active) has another tile
        order = np.lexsort((line_of, inst))
        io, lo, to, co = inst[order], line_of[order], tile[order], counted[order]
        # (uncounted crossings left of the rectangle still occupy a lane; treat them as inactive)
        io, lo, to = io[co], lo[co], to[co]
        head = np.ones(io.size, bool)
        head[1:] = ~((io[1:] == io[:-1]) & (lo[1:] == lo[:-1] + 1) & (to[1:] == to[:-1]))
        req_now = np.unique(np.stack([io[head], to[head] >> line_shift]), axis=1).shape[1]
        atom_now = int(head.sum())
        bo = order[bump_ok[order]]
        ib, lb, tb, db = inst[bo], line_of[bo], btile[bo], delta[bo]
        bhead = np.ones(ib.size, bool)
        bhead[1:] = ~((ib[1:] == ib[:-1]) & (lb[1:] == lb[:-1] + 1) & (tb[1:] == tb[:-1]))
        run_id = np.cumsum(bhead) - 1
        pos_in_run = np.arange(ib.size) - np.flatnonzero(bhead)[run_id]
        pair = run_id * 4096 + pos_in_run // 2
        _, inv = np.unique(pair, return_inverse=True)
        pair_sum = np.bincount(inv, weights=db)
        pair_first = np.zeros(pair_sum.size, np.int64)
        pair_first[inv[::-1]] = np.arange(ib.size)[::-1]
        nz = pair_sum != 0
        breq_now = np.unique(np.stack([ib[pair_first[nz]], tb[pair_first[nz]] >> line_shift]), axis=1).shape[1]
        # aggregated per workgroup chunk
        chunk = line_of // chunk_lines
        ct = np.unique(np.stack([chunk[counted], tile[counted]]), axis=1)
        tiles_agg = ct.shape[1]
        req_agg = np.unique(np.stack([ct[0], ct[1] >> line_shift]), axis=1).shape[1]
        kb = chunk[bump_ok] * (1 << 40) + btile[bump_ok]
        ub, invb = np.unique(kb, return_inverse=True)
        sums = np.bincount(invb, weights=delta[bump_ok])
        ubn = ub[sums != 0]
        breq_agg = np.unique((ubn >> 40) * (1 << 40) + ((ubn & ((1 << 40) - 1)) >> line_shift)).size
        t = 1e6 / 2.7e10
        print(f"  {name}: now {atom_now} count atomics in {req_now} requests + {int(nz.sum())} backdrop atomics in {breq_now} requests = "
              f"{(req_now + breq_now) * t:.0f} us at 2.7e10/s;  per chunk of {chunk_lines}: {tiles_agg} tiles in {req_agg} requests + "
              f"{ubn.size} nonzero backdrop sums in {breq_agg} requests = {(req_agg + breq_agg) * t:.0f} us")


if __name__ == "__main__":
    main()
