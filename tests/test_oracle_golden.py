"""Pins the CPU oracle (and the host encoding mirror) to every in-tree golden the reference holds for
this path.  CPU only."""
import hashlib
import os

import numpy as np
import pytest

import workloads
from oracle import oracle as O
from vello_amd import Affine, Circle, Color, Fill, Rect, Scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLACK = 0xFF000000


def render(scene, w, h, aa=0, base=BLACK):
    packed, layout = scene.resolve()
    o = O.Oracle()
    o.set_scene(packed, layout, w, h, base, aa)
    return o.render(), o


def test_smoke_filled_circle_golden(built):
    # vello_tests/tests/smoke_snapshots.rs:32-48 vs vello_tests/snapshots/smoke/filled_circle.png
    gold = np.load(os.path.join(GOLD, "smoke_goldens.npz"))["filled_circle"]
    img, o = render(workloads.smoke_circle_scene(), 20, 20)
    assert np.abs(img[:, :, :3].astype(int) - gold.astype(int)).max() <= 1
    assert (img[:, :, 3] == 255).all()
    # SURVEY appendix F intermediates
    b = o.bump()
    assert (b["lines"], b["tile"], b["seg_counts"]) == (12, 4, 16)
    assert list(o.buffer("path_bboxes", np.int32)[:4]) == [3, 3, 17, 17]


def test_smoke_filled_square_golden(built):
    gold = np.load(os.path.join(GOLD, "smoke_goldens.npz"))["filled_square"]
    img, _ = render(workloads.smoke_square_scene(), 20, 20)
    assert np.array_equal(img[:, :, :3], gold)


def test_property_simple_square(built):
    # vello_tests/tests/property.rs:21-54: 150x150, 50x50 red square centred at (100,100) -> exactly 2500 red px
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(255, 0, 0), None, Rect.from_center_size((100.0, 100.0), (50.0, 50.0)))
    for aa in (0, 1, 2):
        img, _ = render(s, 150, 150, aa)
        red = (img == np.array([255, 0, 0, 255], dtype=np.uint8)).all(axis=2)
        black = (img == np.array([0, 0, 0, 255], dtype=np.uint8)).all(axis=2)
        assert red.sum() == 2500 and (red | black).all()


def test_property_empty_scene(built):
    # property.rs:56-77: an empty scene is the base colour everywhere
    img, _ = render(Scene(), 33, 17, 0, base=0xFF336699)
    assert (img == np.array([0x99, 0x66, 0x33, 0xFF], dtype=np.uint8)).all()


def test_regression_many_bins(built):
    # vello_tests/tests/regression.rs:213-254: 17x17 bins all red (scaled down from 4352^2 to keep CPU time low:
    # same >256-bin code path needs width_in_bins*height_in_bins > 256 -> 17x17 bins = 4352 px; use 16x17 via 4096x4352?)
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(255, 0, 0), None, Rect(0.0, 0.0, 4352.0, 4352.0))
    img, o = render(s, 4352, 4352, 0)
    assert (img == np.array([255, 0, 0, 255], dtype=np.uint8)).all()


def test_mask_lut_known_answers(built):
    # SURVEY appendix I (f64 transcription of vello_encoding/src/mask.rs:36-98)
    l8, l16 = O.make_mask_lut(), O.make_mask_lut_16()
    assert hashlib.sha256(l8.tobytes()).hexdigest() == "e0a3abedb53b4c28c7f99a0b317afb0f261b491470ddb6599e53a3febf9ee71b"
    assert hashlib.sha256(l16.tobytes()).hexdigest() == "d4a97b10047620de4350610bf3911c013649ccfd8085060ceab3317d0534d61b"
    assert list(l8[:16]) == [0, 0, 8, 8, 8, 8, 72, 72, 72, 72, 74, 74, 74, 74, 106, 106]
    assert list(l16.view(np.uint16)[2048:2056]) == [0xFFFF, 0xFFFF, 0xFEFF, 0xFEFF, 0xFEFF, 0xFEFF, 0xFEFF, 0xFEFE]


def test_watertight_line_soup(built):
    # vello/src/debug/validate.rs:47-64: inside a path every line endpoint appears an even number of times
    for scene in (workloads.circle_scene(), workloads.random_test_scene(3, 60, 256.0, strokes=True)):
        packed, layout = scene.resolve()
        o = O.Oracle()
        o.set_scene(packed, layout, 256, 256, BLACK, 0)
        o.run("pathtag_scan", "flatten")
        n = o.bump()["lines"]
        lines = o.buffer("lines", np.uint32)[: n * 6].reshape(-1, 6)
        for path_ix in np.unique(lines[:, 0]):
            sel = lines[lines[:, 0] == path_ix]
            pts = np.concatenate([sel[:, 2:4], sel[:, 4:6]])
            pts = pts[(sel[:, 2:4] != sel[:, 4:6]).any(axis=1).repeat(2) if False else slice(None)]
            _, counts = np.unique(pts, axis=0, return_counts=True)
            assert (counts % 2 == 0).all(), f"path {path_ix} is not watertight"


def test_tiger_fixture_renders(built):
    d = np.load(os.path.join(GOLD, "tiger_scene.npz"))
    from vello_amd import Layout

    o = O.Oracle()
    o.set_scene(d["packed"], Layout(*[int(v) for v in d["layout"]]), 256, 256, 0xFFFFFFFF, 1)
    img = o.render()
    b = o.bump()
    assert b["failed"] == 0 and b["lines"] > 5000
    assert img.std() > 5  # something was drawn


def _render_resolved(scene, w, h, aa=0, base=BLACK):
    import vello_amd

    r = vello_amd.Resolver().resolve(scene)
    o = O.Oracle()
    o.set_scene(r.packed, r.layout, w, h, base, aa)
    o.set_ramps(r.ramps)
    o.set_image_atlas(r.atlas_image())
    return o.render()


@pytest.mark.parametrize("space", ["premultiplied", "unpremultiplied"])
def test_smoke_gradient_color_alpha_goldens(built, space):
    # vello_tests/tests/regression.rs:150-209 vs snapshots/smoke/gradient_color_alpha_{premultiplied,unpremultiplied}.png:
    # linear gradient (255,255,0,0) -> (0,0,255,255) over a white base, both interpolation alpha spaces.  Pins the
    # ramp generation (ramp_cache.rs), CMD_LIN_GRAD in fine and the src-over onto the base colour.  The reference
    # accepts a mean nv-flip error < 0.001; the oracle reproduces the snapshot exactly.
    gold = np.load(os.path.join(GOLD, "smoke_goldens.npz"))[f"gradient_color_alpha_{space}"]
    img = _render_resolved(workloads.smoke_gradient_alpha_scene(space == "premultiplied"), 100, 50, 0, base=0xFFFFFFFF)
    assert np.array_equal(img[:, :, :3], gold), f"max diff {np.abs(img[:, :, :3].astype(int) - gold.astype(int)).max()}"
    assert (img[:, :, 3] == 255).all()


@pytest.mark.parametrize("extend", ["Pad", "Reflect", "Repeat"])
def test_smoke_data_image_roundtrip_golden(built, extend):
    # regression.rs:33-104 vs snapshots/smoke/data_image_roundtrip.png: the snapshot drawn as an image (nearest
    # sampling, each extend mode) at identity must reproduce itself.  Pins the atlas upload, CMD_IMAGE and the
    # unpremultiplied RGBA8 output.
    from vello_amd import Extend

    gold = np.load(os.path.join(GOLD, "smoke_goldens.npz"))
    rgba, rgb = gold["data_image_roundtrip_rgba"], gold["data_image_roundtrip_rgb"]
    h, w = rgba.shape[:2]
    img = _render_resolved(workloads.smoke_data_image_scene(rgba, getattr(Extend, extend)), w, h, 0)
    assert np.array_equal(img[:, :, :3], rgb), f"max diff {np.abs(img[:, :, :3].astype(int) - rgb.astype(int)).max()}"


def _premultiplied_distance(img, want):
    """color::PremulColor::difference: the Euclidean distance of the premultiplied components"""
    got = img.reshape(-1, 4).astype(np.float64) / 255.0
    got[:, :3] *= got[:, 3:4]
    return np.sqrt(((got - want) ** 2).sum(axis=1))


def test_property_bgra_image(built):
    # vello_tests/tests/property.rs:107-149: Bgra8 bytes come out as the colours they encode (pixel_format, fine.wgsl:837-851),
    # each within 1e-4 of RED, BLUE, LIME, WHITE (premultiplied distance); the target is 2 x 2 over the default black.
    scene, want = workloads.property_image_scene("bgra")
    img = _render_resolved(scene, 2, 2, 0, base=BLACK)
    assert (_premultiplied_distance(img, want) <= 1e-4).all(), img.reshape(-1, 4)


def test_property_premultiplied_image(built):
    # property.rs:151-199: premultiplied Rgba8 bytes (alpha 0.5) over a transparent base come out un-premultiplied such that
    # premultiplying them again is within 1e-2 of what went in (maybe_premul_alpha, fine.wgsl:853-863; the output's
    # un-premultiplication, fine.wgsl:1386-1397)
    scene, want = workloads.property_image_scene("premultiplied")
    img = _render_resolved(scene, 2, 2, 0, base=0x00000000)
    assert (_premultiplied_distance(img, want) <= 1e-2).all(), img.reshape(-1, 4)


def test_oracle_pools_grow_on_overflow():
    """a frame that overflows a pool doubles the oracle's pools and runs again (auto_grow), so scenes beyond
    any fixed pool are still checked by something; without auto_grow the overflow is reported as before."""
    import workloads
    from oracle.oracle import Oracle

    scene = workloads.random_test_scene(3, n_paths=400, size=512.0, strokes=True, clips=False)
    packed, layout = scene.resolve()
    # an 8192^2 target is 512 x 512 tiles x 64 words = 2^24 words of fixed PTCL blocks: beyond config.rs:408's 2^23
    fixed = Oracle(capacity_scale=1)
    fixed.set_scene(packed, layout, 8192, 8192, 0xFF000000, 0)
    fixed.render()
    assert fixed.bump()["failed"] != 0 and fixed.grown == 0
    o = Oracle(capacity_scale=1, auto_grow=True, max_capacity_scale=64)
    o.set_scene(packed, layout, 8192, 8192, 0xFF000000, 0)
    img = o.render()
    assert o.grown >= 1 and o.capacity_scale() >= 2 and o.bump()["failed"] == 0
    # the same frame from pools that were large enough from the start
    big = Oracle(capacity_scale=4)
    big.set_scene(packed, layout, 8192, 8192, 0xFF000000, 0)
    assert (big.render() == img).all() and big.bump()["failed"] == 0
