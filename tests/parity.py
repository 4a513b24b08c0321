"""Shared comparison helpers: HIP engine (real or emulated) vs the CPU oracle, stage by stage."""
import os

import numpy as np

from oracle.oracle import Oracle

BUMP_KEYS = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]
DUMP_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_dumps")


def _dump(name, **arrays):
    try:
        os.makedirs(DUMP_DIR, exist_ok=True)
        np.savez_compressed(os.path.join(DUMP_DIR, name + ".npz"), **arrays)
    except OSError:
        pass


def sorted_rows(a, ncol):
    a = np.ascontiguousarray(a).reshape(-1, ncol)
    order = np.lexsort(a.T[::-1])
    return a[order]


def canonical_nan_words(words):
    """u32 words with every f32-NaN bit pattern replaced by ONE quiet NaN (see canonical_nan_lines); applied to both
    sides, so words that are not floats stay comparable."""
    w = np.ascontiguousarray(words).copy()
    w[((w & 0x7F800000) == 0x7F800000) & ((w & 0x007FFFFF) != 0)] = 0x7FC00000
    return w


def canonical_nan_lines(words):
    """LineSoup rows (u32 x 6) with every NaN coordinate replaced by ONE quiet-NaN pattern: which sign / payload a NaN
    carries out of an arithmetic operation is implementation-defined (x86 and gfx950 differ), and a NaN line is a NaN
    line -- it crosses no tile (scenes reach this only through garbage transforms, e.g. a path encoded before any
    transform exists)."""
    w = np.ascontiguousarray(words).reshape(-1, 6).copy()
    f = w[:, 2:]
    f[((f & 0x7F800000) == 0x7F800000) & ((f & 0x007FFFFF) != 0)] = 0x7FC00000
    return w


def fine_on_engine_inputs(engine, oracle, width, height, used_words):
    """The oracle's fine stage run on the ENGINE's fine inputs (segments in the engine's order, its PTCL, tiles and
    draw info): isolates fine from the one nondeterminism of the path, the order atomics give the segments of a tile
    (area AA sums their f32 contributions in that order; fine.wgsl:1020-1066)."""
    for name in ("segments", "ptcl", "tiles", "info_bin_data"):
        dst = oracle.buffer(name, np.uint8)
        src = engine.read_buffer(name, np.uint8, min(dst.size, used_words[name] * 4))
        dst[: src.size] = src
    oracle.run("fine", "fine")
    return oracle.buffer("output", np.uint8)[: width * height * 4].reshape(height, width, 4).copy()


# ---------------------------------------------------------------------------------------------------------------
# Back half of the pipeline: bin lists, SegmentCount records, PTCL words, per-tile segment slices.
# What the order of atomics decides (and the reference's GPUs decide differently from run to run as well) is
# normalised away; everything else must be equal word for word.
# ---------------------------------------------------------------------------------------------------------------
CMD_END, CMD_FILL, CMD_SOLID, CMD_COLOR, CMD_LIN_GRAD, CMD_RAD_GRAD, CMD_SWEEP_GRAD = 0, 1, 3, 5, 6, 7, 8
CMD_IMAGE, CMD_BEGIN_CLIP, CMD_END_CLIP, CMD_JUMP, CMD_BLUR_RECT = 9, 10, 11, 12, 13
_CMD_SIZE = np.zeros(16, dtype=np.int64)
for _t, _n in ((CMD_FILL, 4), (CMD_SOLID, 1), (CMD_COLOR, 2), (CMD_LIN_GRAD, 3), (CMD_RAD_GRAD, 3), (CMD_SWEEP_GRAD, 3), (CMD_IMAGE, 2),
               (CMD_BEGIN_CLIP, 1), (CMD_END_CLIP, 3), (CMD_BLUR_RECT, 3)):
    _CMD_SIZE[_t] = _n


def compare_bins(name, bh_h, bd_h, bh_o, bd_o, n_draw, width, height, bin_data_start):
    """binning.wgsl:55-203: per (partition of 256 draw objects, bin) the element count and the element list, in order
    (the rank of an element inside its chunk is a popcount, not an atomic); chunk_offset itself follows bump order."""
    wb, hb = (((width + 15) // 16) + 15) // 16, (((height + 15) // 16) + 15) // 16
    n_bins = wb * hb
    aligned = (n_bins + 255) // 256 * 256
    n_part = (n_draw + 255) // 256
    n_cmp = 0
    for part in range(n_part):
        hh = bh_h[part * aligned * 2: (part * aligned + n_bins) * 2].reshape(-1, 2)
        ho = bh_o[part * aligned * 2: (part * aligned + n_bins) * 2].reshape(-1, 2)
        assert np.array_equal(hh[:, 0], ho[:, 0]), f"{name}: bin_headers element counts differ in partition {part}"
        for b in np.nonzero(ho[:, 0])[0]:
            n = int(ho[b, 0])
            lh = bd_h[bin_data_start + int(hh[b, 1]): bin_data_start + int(hh[b, 1]) + n]
            lo = bd_o[bin_data_start + int(ho[b, 1]): bin_data_start + int(ho[b, 1]) + n]
            assert np.array_equal(lh, lo), f"{name}: bin_data differs (partition {part}, bin {b})"
            n_cmp += n
    return n_cmp


def compare_seg_counts(name, sc_h, lines_h, sc_o, lines_o, n):
    """path_count.wgsl:172-199: one (line, crossing index within the line) record per tile crossing.  line_ix follows the
    order of the line soup and the slot within the tile's slice follows the tile atomics: records are compared as the
    multiset of (line CONTENT, seg_within_line); the slots are checked through the per-tile segment slices."""
    def canon(sc, lines):
        sc = sc[: n * 2].reshape(-1, 2)
        rows = canonical_nan_lines(lines)[sc[:, 0]]
        return sorted_rows(np.concatenate([rows, (sc[:, 1] & 0xffff)[:, None]], axis=1), 7)
    a, b = canon(sc_h, lines_h), canon(sc_o, lines_o)
    assert np.array_equal(a, b), f"{name}: seg_counts differ as a multiset of (line, crossing) ({(a != b).any(axis=1).sum()} rows)"


def walk_ptcl_pair(name, ptcl_h, ptcl_o, n_tiles):
    """Walks the command lists of all tiles of both PTCL buffers in lockstep (vectorised over tiles).  Asserts the
    streams are equal word for word except (a) CMD_JUMP targets / chunk placement (bump order), (b) the segment index
    of CMD_FILL (bump order) and (c) word 0 of a tile, its blend-spill offset (bump order).  Returns the fills as
    (seg_ix_hip, seg_ix_oracle, n_segs) arrays and the number of words compared."""
    tiles = np.arange(n_tiles, dtype=np.int64)
    ih, io = tiles * 64 + 1, tiles * 64 + 1
    active = np.ones(n_tiles, dtype=bool)
    fills = []
    n_words = 0

    def follow(ptcl, ix, act):
        for _ in range(1 << 20):
            j = act & (ptcl[np.minimum(ix, ptcl.size - 1)] == CMD_JUMP)
            if not j.any():
                return ix
            ix = np.where(j, ptcl[np.minimum(ix + 1, ptcl.size - 1)].astype(np.int64), ix)
        raise AssertionError(f"{name}: PTCL jump cycle")

    for _ in range(1 << 22):
        if not active.any():
            break
        ih, io = follow(ptcl_h, ih, active), follow(ptcl_o, io, active)
        a = np.nonzero(active)[0]
        th, to = ptcl_h[ih[a]], ptcl_o[io[a]]
        if not np.array_equal(th, to):
            bad = a[np.nonzero(th != to)[0][0]]
            raise AssertionError(f"{name}: PTCL command differs in tile {bad}: hip {ptcl_h[ih[bad]]} oracle {ptcl_o[io[bad]]}")
        assert (th < 16).all() and ((th == CMD_END) | (_CMD_SIZE[np.minimum(th, 15)] > 0)).all(), f"{name}: unknown PTCL command"
        size = _CMD_SIZE[th]
        for k in (1, 2, 3):
            m = size > k
            if not m.any():
                continue
            wh, wo = ptcl_h[ih[a[m]] + k], ptcl_o[io[a[m]] + k]
            if k == 2:  # CMD_FILL word 2 = segment index: allocation order
                isf = th[m] == CMD_FILL
                fills.append((wh[isf].astype(np.int64), wo[isf].astype(np.int64), (ptcl_o[io[a[m]][isf] + 1] >> 1).astype(np.int64)))
                wh, wo = wh[~isf], wo[~isf]
            if not np.array_equal(wh, wo):
                raise AssertionError(f"{name}: PTCL payload word {k} differs ({(wh != wo).sum()} commands)")
        n_words += int(size.sum()) + int((th == CMD_END).sum())
        ih[a] += size
        io[a] += size
        active[a[th == CMD_END]] = False
    assert not active.any(), f"{name}: PTCL walk did not terminate"
    if fills:
        fh, fo, fn = (np.concatenate(x) for x in zip(*fills))
    else:
        fh = fo = fn = np.zeros(0, dtype=np.int64)
    return fh, fo, fn, n_words


def compare_segment_slices(name, seg_h, seg_o, fh, fo, fn):
    """path_tiling.wgsl:39-173: the PathSegment slice of every CMD_FILL, as a multiset per fill (the slot of a segment
    inside its tile's slice is the order of the tile atomic in path_count)."""
    total = int(fn.sum())
    if total == 0:
        return 0
    fill_id = np.repeat(np.arange(fn.size, dtype=np.int64), fn)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(fn) - fn, fn)
    rows_h = canonical_nan_words(seg_h.reshape(-1, 6)[np.repeat(fh, fn) + within])
    rows_o = canonical_nan_words(seg_o.reshape(-1, 6)[np.repeat(fo, fn) + within])

    def canon(rows):
        key = np.concatenate([fill_id[:, None].astype(np.uint64), rows.astype(np.uint64)], axis=1)
        return key[np.lexsort(key.T[::-1])]
    a, b = canon(rows_h), canon(rows_o)
    assert np.array_equal(a, b), f"{name}: segments differ within {len(np.unique(a[(a != b).any(axis=1), 0]))} tile slices"
    return total


def compare_back_half(engine, oracle, packed, layout, width, height, base_color, aa, name, ramps=None, ref=None):
    """Renders once more with occlusion culling off (VELLO_HIP_DEBUG_NO_CULL): every counter must then equal the
    oracle's, and bin lists, SegmentCounts, PTCL and segment slices are diffed."""
    engine.update_debug_flags(no_cull=True)
    try:
        img, bump = engine.render(packed, layout, width, height, base_color, aa, ramps=ramps)
    finally:
        engine.update_debug_flags(no_cull=False)
    ob = oracle.bump()
    # bump.ptcl is the one counter that is not the reference's: the engine stores a tile's commands in exact-fit regions
    # linked by CMD_JUMP instead of 256-word chunks (k_coarse); the command words themselves are diffed below
    assert all(bump[k] == ob[k] for k in BUMP_KEYS if k != "ptcl"), f"{name}: bump counters differ with culling off: hip {bump} oracle {ob}"
    if bump["failed"] != 0:
        return {}
    L = layout
    n_tiles = ((width + 15) // 16) * ((height + 15) // 16)
    stats = {}
    bh_h = engine.read_buffer("bin_headers", np.uint32)
    bd_h = engine.read_buffer("info_bin_data", np.uint32, (L.bin_data_start + ob["binning"]) * 4)
    stats["bin_entries"] = compare_bins(name, bh_h, bd_h, oracle.buffer("bin_headers", np.uint32), oracle.buffer("info_bin_data", np.uint32),
                                        L.n_draw_objects, width, height, L.bin_data_start)
    assert stats["bin_entries"] == ob["binning"], f"{name}: bin lists hold {stats['bin_entries']} entries, bump.binning {ob['binning']}"
    n_lines, n_sc = ob["lines"], ob["seg_counts"]
    compare_seg_counts(name, engine.read_buffer("seg_counts", np.uint32, n_sc * 8), engine.read_buffer("lines", np.uint32, n_lines * 24),
                       oracle.buffer("seg_counts", np.uint32), oracle.buffer("lines", np.uint32)[: n_lines * 6], n_sc)
    stats["seg_counts"] = n_sc
    ptcl_h = engine.read_buffer("ptcl", np.uint32, (64 * n_tiles + bump["ptcl"]) * 4)
    ptcl_o = oracle.buffer("ptcl", np.uint32)[:64 * n_tiles + ob["ptcl"]]
    fh, fo, fn, stats["ptcl_words"] = walk_ptcl_pair(name, ptcl_h, ptcl_o, n_tiles)
    assert int(fn.sum()) == ob["segments"], f"{name}: CMD_FILLs cover {int(fn.sum())} segments, bump.segments {ob['segments']}"
    stats["segments"] = compare_segment_slices(name, engine.read_buffer("segments", np.uint32, ob["segments"] * 24),
                                               oracle.buffer("segments", np.uint32)[: ob["segments"] * 6], fh, fo, fn)
    if ref is not None and aa != 0:
        assert np.array_equal(img, ref), f"{name}: image differs with culling off"
    return stats


def compare_frame(engine, packed, layout, width, height, base_color, aa, name, tol=0, check_stages=True, oracle=None,
                  resolved=None, order_sensitive=False, min_agree=0.99, back_half=True):
    """Renders with both, asserts bump counters, intermediates (up to documented permutations) and the
    final RGBA8 image agree.  tol is the per-channel tolerance on the image (0 for MSAA: integer coverage;
    <=1 for area AA where segment order changes f32 summation order).

    For area AA the image is ALSO compared exactly against the oracle's fine run on the engine's own segment order.
    order_sensitive=True (fuzzed scenes) drops the direct +-tol comparison and keeps that exact one: a last-ulp
    difference in a pixel's area can be amplified without bound by what composites it (un-premultiplication at alpha
    ~ 0, Compose modes that divide by alpha, ColorDodge / ColorBurn / the non-separable mix modes), on the reference's
    own GPUs as much as here.  min_agree: the fraction of pixels on which the two orders must still agree within tol
    (None for scenes with non-finite geometry, where a NaN wins or loses a min / max depending on the order)."""
    oracle = oracle or Oracle()
    oracle.set_scene(packed, layout, width, height, base_color, int(aa))
    ramps = None
    if resolved is not None:  # late-bound resources: gradient ramps + image atlas (vello_amd.Resolver)
        ramps = resolved.ramps
        oracle.set_ramps(ramps)
        oracle.set_image_atlas(resolved.atlas_image())
        if resolved.atlas_size:
            engine.resize_image_atlas(resolved.atlas_size, resolved.atlas_size)
            for x, y, px in resolved.uploads:
                engine.write_image(x, y, px)
    ref = oracle.render()
    img, bump = engine.render(packed, layout, width, height, base_color, aa, ramps=ramps)
    ob = oracle.bump()
    # Occlusion culling in coarse (scenes without clips) skips draws hidden under an opaque full-tile cover:
    # the segment / PTCL demand can only shrink; every other counter must match exactly.
    # (a restarted list re-allocates its chunks, so bump.ptcl can land on either side of the oracle's; with culling off
    # -- compare_back_half -- every counter is exact)
    exact = [k for k in BUMP_KEYS if k not in ("segments", "ptcl")]
    ok = all(bump[k] == ob[k] for k in exact) and bump["segments"] <= ob["segments"]
    if layout.n_clips != 0:
        ok = ok and all(bump[k] == ob[k] for k in BUMP_KEYS if k != "ptcl")
    if not ok:
        _dump(name + "_bump", img=img, ref=ref)
    assert ok, f"{name}: bump counters differ: hip {bump} oracle {ob}"
    L = layout
    if check_stages:
        n_tw = (L.path_data_base - L.path_tag_base)
        tm_h = engine.read_buffer("tag_monoids", np.uint32, n_tw * 20)
        tm_o = oracle.buffer("tag_monoids", np.uint32)[: n_tw * 5]
        assert np.array_equal(tm_h, tm_o), f"{name}: tag_monoids differ"
        pb_h = engine.read_buffer("path_bboxes", np.int32, L.n_paths * 24)
        pb_o = oracle.buffer("path_bboxes", np.int32)[: L.n_paths * 6]
        if not np.array_equal(pb_h, pb_o):
            _dump(name + "_path_bboxes", hip=pb_h, oracle=pb_o)
        assert np.array_equal(pb_h, pb_o), f"{name}: path_bboxes differ"
        n_lines = ob["lines"]
        ln_h = sorted_rows(canonical_nan_lines(engine.read_buffer("lines", np.uint32, n_lines * 24)), 6)
        ln_o = sorted_rows(canonical_nan_lines(oracle.buffer("lines", np.uint32)[: n_lines * 6]), 6)
        if not np.array_equal(ln_h, ln_o):
            _dump(name + "_lines", hip=ln_h, oracle=ln_o)
        assert np.array_equal(ln_h, ln_o), f"{name}: line soup differs as a multiset ({(ln_h != ln_o).any(axis=1).sum()} rows)"
        dm_h = engine.read_buffer("draw_monoids", np.uint32, L.n_draw_objects * 16)
        dm_o = oracle.buffer("draw_monoids", np.uint32)[: L.n_draw_objects * 4]
        assert np.array_equal(dm_h, dm_o), f"{name}: draw_monoids differ"
        info_h = engine.read_buffer("info_bin_data", np.uint32, L.bin_data_start * 4)
        info_o = oracle.buffer("info_bin_data", np.uint32)[: L.bin_data_start]
        assert np.array_equal(canonical_nan_words(info_h), canonical_nan_words(info_o)), f"{name}: draw info differs"
        if L.n_clips:
            cb_h = engine.read_buffer("clip_bboxes", np.uint32, L.n_clips * 16)
            cb_o = oracle.buffer("clip_bboxes", np.uint32)[: L.n_clips * 4]
            assert np.array_equal(cb_h, cb_o), f"{name}: clip_bboxes differ"
        db_h = engine.read_buffer("draw_bboxes", np.uint32, L.n_draw_objects * 16)
        db_o = oracle.buffer("draw_bboxes", np.uint32)[: L.n_draw_objects * 4]
        assert np.array_equal(db_h, db_o), f"{name}: draw_bboxes differ"
        # Path records: bbox exact; tile offsets follow bump order -> compare per-path tile contents instead
        p_h = engine.read_buffer("paths", np.uint32, L.n_draw_objects * 32).reshape(-1, 8)
        p_o = oracle.buffer("paths", np.uint32)[: L.n_draw_objects * 8].reshape(-1, 8)
        assert np.array_equal(p_h[:, :4], p_o[:, :4]), f"{name}: path tile bboxes differ"
        t_h = engine.read_buffer("tiles", np.int32, ob["tile"] * 8).reshape(-1, 2)
        t_o = oracle.buffer("tiles", np.int32)[: ob["tile"] * 2].reshape(-1, 2)
        # backdrops are order independent; segment_count_or_ix holds ~index after coarse -> compare backdrops
        bd_ok = True
        for i in range(L.n_draw_objects):
            n = int((p_o[i, 2] - p_o[i, 0]) * (p_o[i, 3] - p_o[i, 1]))
            if n and not np.array_equal(t_h[p_h[i, 4]: p_h[i, 4] + n, 0], t_o[p_o[i, 4]: p_o[i, 4] + n, 0]):
                bd_ok = False
                break
        assert bd_ok, f"{name}: tile backdrops differ (path {i})"
    diff = np.abs(img.astype(np.int32) - ref.astype(np.int32))
    if not order_sensitive:
        if diff.max() > tol:
            _dump(name + "_image", hip=img, oracle=ref)
        assert diff.max() <= tol, f"{name}: image differs from oracle: max {diff.max()}, {(diff > tol).sum()} values over tol {tol}"
    if tol > 0 and bump["failed"] == 0:
        n_tiles = ((width + 15) // 16) * ((height + 15) // 16)
        used = {"segments": bump["segments"] * 6, "ptcl": 64 * n_tiles + bump["ptcl"], "tiles": ob["tile"] * 2,
                "info_bin_data": L.bin_data_start + ob["binning"]}
        same_order = fine_on_engine_inputs(engine, oracle, width, height, used)
        if not np.array_equal(img, same_order):
            _dump(name + "_image_same_order", hip=img, oracle=same_order)
        assert np.array_equal(img, same_order), (
            f"{name}: fine differs from the oracle's fine on the same segment order: "
            f"max {np.abs(img.astype(np.int32) - same_order.astype(np.int32)).max()}")
        if order_sensitive and min_agree is not None:  # still require the two orders to agree almost everywhere
            frac = float((diff.max(axis=2) > tol).mean())
            assert frac <= 1.0 - min_agree, f"{name}: {frac:.2%} of the pixels differ by more than {tol} between the two segment orders"
    # LAST (it renders once more, with culling off, and leaves that frame in the engine's buffers).
    # back_half=False: scenes whose crossing indices leave f32's 24 bits / SegmentCount's 16 bits -- two records can
    # then claim the same slot of a foreign tile and which one stays is the order of the stores
    if check_stages and back_half and bump["failed"] == 0 and ob["failed"] == 0:
        if tol > 0:
            oracle.render()  # the same-order check overwrote the oracle's tiles / PTCL / segments with the engine's
        compare_back_half(engine, oracle, packed, layout, width, height, base_color, aa, name, ramps=ramps, ref=ref)
    return img, ref, bump


def clip_ops_scene(ops, rng):
    """A scene of nothing but clip layers: ops[i] > 0 pushes a layer clipped to a random rectangle, otherwise the innermost
    open layer is popped (if there is one).  Layers still open at the end are closed by resolve (resolve.rs:127-141)."""
    from vello_amd import Affine, Fill, Rect, Scene

    s = Scene()
    depth = 0
    for o in ops:
        if o > 0:
            x0, y0 = rng.uniform(0, 200, 2)
            w, h = rng.uniform(5, 300, 2)
            s.push_clip_layer(Fill.NonZero, Affine.IDENTITY, Rect(x0, y0, x0 + w, y0 + h))
            depth += 1
        elif depth > 0:
            s.pop_layer()
            depth -= 1
    return s


def compare_clip_stage(engine, ops, rng, name, oracle=None):
    """clip_reduce + clip_leaf alone (clip_reduce.wgsl:24-67, clip_leaf.wgsl:80-217): the front stages up to the clip stage run
    on the engine -- once with its partitioned kernels, once with the one-wave stack machine (VELLO_HIP_DEBUG_SEQ_CLIP) -- and
    on the oracle; clip_bboxes and the patched draw monoids must be equal word for word."""
    from vello_amd import AaConfig

    packed, layout = clip_ops_scene(ops, rng).resolve()
    o = oracle or Oracle()
    o.set_scene(packed, layout, 256, 256, 0xFF000000, int(AaConfig.Area))
    o.run(0, 3)
    ref_cb = o.buffer("clip_bboxes", np.float32)[: layout.n_clips * 4].copy()
    ref_dm = o.buffer("draw_monoids", np.uint32)[: layout.n_draw_objects * 4].copy()
    try:
        for flags in ("partitioned kernels", "one-wave stack machine"):
            engine.set_debug_flags(seq_clip=flags == "one-wave stack machine")
            engine.upload_scene(packed, layout)
            engine.run_stages(256, 256, 0xFF000000, AaConfig.Area, 0, 3)
            cb = engine.read_buffer("clip_bboxes", np.float32)[: layout.n_clips * 4]
            dm = engine.read_buffer("draw_monoids", np.uint32)[: layout.n_draw_objects * 4]
            bad = np.nonzero(cb != ref_cb)[0]
            assert bad.size == 0, f"{name} ({flags}): clip_bboxes differ first at clip {bad[0] // 4}: {cb[bad[:4]]} vs {ref_cb[bad[:4]]}"
            bad = np.nonzero(dm != ref_dm)[0]
            assert bad.size == 0, f"{name} ({flags}): draw_monoids differ first at draw object {bad[0] // 4}"
    finally:
        engine.set_debug_flags()
    return layout


def clip_structures(big):
    """(name, ops) pairs for compare_clip_stage: +1 pushes a clip layer, -1 pops.  The partitioned kernels cut the clip stream every
    256 clips: runs, teeth and depths are chosen around that, the stack grows far beyond the 256 entries clip_leaf.wgsl:87-112 holds."""
    yield "one", [1, -1]
    yield "flat", [1, -1] * 300
    yield "deep", [1] * 700 + [-1] * 700
    yield "outer+flat", [1] * 300 + [1, -1] * 400 + [-1] * 300
    yield "left open", [1] * 5 + [1, -1] * 200 + [1] * 600
    for seed in range(12):
        rng = np.random.default_rng(100 + seed)
        n = int(rng.integers(1, 4000))
        if seed % 4 == 0:
            ops = np.where(rng.random(n) < rng.uniform(0.3, 0.7), 1, -1)
        elif seed % 4 == 1:  # long runs
            ops = np.repeat(np.where(rng.random(n // 50 + 1) < 0.5, 1, -1), rng.integers(1, 400, n // 50 + 1))[:n]
        elif seed % 4 == 2:  # teeth around the partition size
            ops = np.concatenate([[1] * int(rng.integers(200, 300)) + [-1] * int(rng.integers(100, 300)) for _ in range(n // 400 + 1)])
        else:
            ops = np.where(rng.random(n) < 0.5 + 0.3 * np.sin(np.arange(n) / 97.0), 1, -1)
        yield f"random {seed}", list(ops)
    if big:
        yield "20 000 deep", [1] * 20000 + [-1] * 20000
        n = 300000  # > 1024 partitions: two per thread in k_clip_stack
        yield "300 000 clips", list(np.where(np.random.default_rng(5).random(n) < 0.5 + 0.2 * np.sin(np.arange(n) / 3001.0), 1, -1))


def front_fusion_cases():
    """Small scenes whose stages share launches (k_front, flatten.hip): (name, packed, layout, w, h, background, launches per frame)
    -- 1: everything up to tile_alloc as one workgroup's work (a few dozen segments, no clips); 2: [zero fill | pathtag scan | light
    pass + draw scan] and [binning | tile_alloc]."""
    import workloads
    from vello_amd import Layout, Scene

    out = []
    s = workloads.smoke_circle_scene()
    out.append(("circle",) + tuple(s.resolve()) + (20, 20, 0xFF000000, 1))
    s = workloads.circle_scene()
    out.append(("circle256",) + tuple(s.resolve()) + (256, 256, 0xFF000000, 1))
    out.append(("empty",) + tuple(Scene().resolve()) + (64, 64, 0xFF102030, 1))
    out.append(("stroke_styles",) + tuple(workloads.stroke_styles_scene().resolve()) + (256, 256, 0xFFFFFFFF, 2))
    out.append(("clip_blend",) + tuple(workloads.clip_blend_scene().resolve()) + (256, 256, 0xFF000000, 2))
    out.append(("random_700",) + tuple(workloads.random_test_scene(3, n_paths=700, size=384.0, strokes=True, clips=True).resolve()) + (384, 384, 0xFF000000, 2))
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiger_scene.npz"))
    out.append(("tiger", d["packed"], Layout(*[int(v) for v in d["layout"]]), 320, 320, 0xFFFFFFFF, 2))
    return out


def check_front_fusion(engine, case, in_flight=(1, 2)):
    """One of front_fusion_cases(): the fused launches taken (vello_hip_fused_launches) and every stage and the image against the
    oracle; then the same scene with every stage as a kernel of its own (VELLO_HIP_DEBUG_NO_FUSION)."""
    from vello_amd import AaConfig

    name, packed, layout, w, h, bg, per_frame = case
    try:
        # The launch counts asserted below depend on which of flatten's kernel sets the engine picks (the one-launch front needs the
        # cooperative set) and the engine picks it from the scene's size and the list counts of its last finished frame: pinned
        # here, so that the counts do not depend on what the engine rendered before.  (Stage profiling --
        # vello_hip_set_profiling -- also decides fusion: a stage that is timed on its own is launched on its own.)
        engine.set_debug_flags(flatten_coop=True)
        for n in in_flight:
            engine.set_frames_in_flight(n)
            for aa in (AaConfig.Area, AaConfig.Msaa16):
                before = engine.fused_launches()
                engine.render(packed, layout, w, h, bg, aa)
                assert engine.fused_launches() - before == per_frame, (name, n, engine.fused_launches() - before, per_frame)
                compare_frame(engine, packed, layout, w, h, bg, aa, f"fusion_{name}_{n}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0)
        engine.set_frames_in_flight(1)
        engine.set_debug_flags(no_fusion=True)
        before = engine.fused_launches()
        compare_frame(engine, packed, layout, w, h, bg, AaConfig.Msaa16, f"nofusion_{name}")
        assert engine.fused_launches() == before
    finally:
        engine.set_frames_in_flight(1)
        engine.set_debug_flags()
