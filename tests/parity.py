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


def compare_frame(engine, packed, layout, width, height, base_color, aa, name, tol=0, check_stages=True, oracle=None,
                  resolved=None, order_sensitive=False, min_agree=0.99):
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
    exact = [k for k in BUMP_KEYS if k not in ("segments", "ptcl")]
    ok = all(bump[k] == ob[k] for k in exact) and bump["segments"] <= ob["segments"] and bump["ptcl"] <= ob["ptcl"]
    if layout.n_clips != 0:
        ok = ok and bump == ob
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
    return img, ref, bump
