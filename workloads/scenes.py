"""Seeded synthetic scenes.  Randomness: numpy PCG64 (np.random.Generator) with the seeds named in
SURVEY.md 8d d2; the generators are the definition of the workload (no network, no datasets)."""
import math

import numpy as np

from vello_amd import Affine, BezPath, Cap, Circle, Color, Compose, BlendMode, Fill, Join, Mix, Rect, Scene, Stroke
from vello_amd.kurbo import MOVE_TO, LINE_TO, QUAD_TO, CURVE_TO, CLOSE_PATH

PALETTE = [Color.from_rgb8(*c) for c in [
    (0xf2, 0x8c, 0xa8), (0x26, 0x46, 0x53), (0x2a, 0x9d, 0x8f), (0xe9, 0xc4, 0x6a), (0xf4, 0xa2, 0x61), (0xe7, 0x6f, 0x51),
    (0x8e, 0xca, 0xe6), (0x21, 0x9e, 0xbc), (0x02, 0x30, 0x47), (0xff, 0xb7, 0x03), (0xfb, 0x85, 0x00), (0x60, 0x6c, 0x38),
    (0x28, 0x36, 0x18), (0xdd, 0xa1, 0x5e), (0xbc, 0x6c, 0x25), (0x6d, 0x59, 0x7a)]]


def circle_scene():
    """C1: Circle((128,128), r=100), NonZero, README colour rgb8(242,140,168); 256x256."""
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(242, 140, 168), None, Circle((128.0, 128.0), 100.0))
    return s


def smoke_circle_scene():
    """vello_tests/tests/smoke_snapshots.rs:32-48."""
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(0, 0, 255), None, Circle((10.0, 10.0), 7.0))
    return s


def smoke_square_scene():
    """vello_tests/tests/smoke_snapshots.rs:14-30."""
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(0, 0, 255), None, Rect.from_center_size((10.0, 10.0), (6.0, 6.0)))
    return s


def smoke_gradient_alpha_scene(premultiplied):
    """vello_tests/tests/regression.rs:150-209 (test_gradient_color_alpha_{premultiplied,unpremultiplied}): 100x50,
    base colour white."""
    from vello_amd import Gradient, InterpolationAlphaSpace
    sp = InterpolationAlphaSpace.Premultiplied if premultiplied else InterpolationAlphaSpace.Unpremultiplied
    g = Gradient.new_linear((0.0, 0.0), (100.0, 0.0)).with_stops(
        [(0.0, Color.from_rgba8(255, 255, 0, 0)), (1.0, Color.from_rgba8(0, 0, 255, 255))]).with_interpolation_alpha_space(sp)
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(0.0, 0.0, 100.0, 50.0))
    return s


def smoke_data_image_scene(rgba, extend):
    """vello_tests/tests/regression.rs:33-104 (test_data_image_roundtrip_extend_{pad,reflect,repeat}): the image drawn at
    identity with nearest sampling into a target of its own size."""
    from vello_amd import ImageBrush, ImageData, ImageQuality
    s = Scene()
    s.draw_image(ImageBrush(ImageData(rgba), extend, extend, ImageQuality.Low), Affine.IDENTITY)
    return s


def property_image_scene(kind):
    """vello_tests/tests/property.rs:107-199: a 2 x 2 image of RED, BLUE, LIME, WHITE drawn at identity with nearest sampling --
    `bgra_image` (Bgra8 bytes, straight alpha) and `premultiplied_image` (Rgba8, alpha 0.5, premultiplied bytes).  Returns the
    scene and the four colours as the tests compare them (rgba floats, premultiplied)."""
    from vello_amd import ImageAlphaType, ImageBrush, ImageData, ImageFormat, ImageQuality
    rgba = np.array([[255, 0, 0, 255], [0, 0, 255, 255], [0, 255, 0, 255], [255, 255, 255, 255]], dtype=np.float64) / 255.0
    if kind == "bgra":
        px = np.round(rgba[:, [2, 1, 0, 3]] * 255.0).astype(np.uint8).reshape(2, 2, 4)
        image = ImageData(px, ImageFormat.Bgra8, ImageAlphaType.Alpha)
        want = rgba.copy()
        want[:, :3] *= want[:, 3:4]
    else:
        pm = rgba.copy()
        pm[:, 3] = 0.5
        pm[:, :3] *= 0.5  # palette colour .with_alpha(0.5).premultiply()
        # PremulColor::to_rgba8: round(255 x)
        px = np.floor(pm * 255.0 + 0.5).astype(np.uint8).reshape(2, 2, 4)
        image = ImageData(px, ImageFormat.Rgba8, ImageAlphaType.AlphaPremultiplied)
        want = pm
    s = Scene()
    s.draw_image(ImageBrush(image, quality=ImageQuality.Low), Affine.IDENTITY)
    return s, want


def _polyline(pts, closed):
    n = len(pts)
    verbs = np.full(n + (1 if closed else 0), LINE_TO, dtype=np.uint8)
    verbs[0] = MOVE_TO
    if closed:
        verbs[-1] = CLOSE_PATH
    return BezPath.from_arrays(verbs, np.asarray(pts, dtype=np.float64).reshape(-1))


def paris_like_scene(seed=0x5EED0001, n_paths=30000, size=1600.0, stroke_frac=0.16, blob_frac=0.03):
    """C3: SYNTHETIC stand-in for paris-30k (a city map: roads as thin stroked polylines, buildings/parks as
    filled polygons, a few curved blobs).  Like the real asset it is made of many SHORT segments (30k paths,
    ~1.1-1.2 M path tags, ~10 MB encoding at f32) and is sized to fit the reference's own fixed pool
    capacities (2^21 lines / crossings / tiles, vello_encoding/src/config.rs:401-408)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    s = Scene()
    ident = Affine.IDENTITY
    kinds = rng.random(n_paths)
    for i in range(n_paths):
        col = PALETTE[int(rng.integers(0, len(PALETTE)))]
        k = kinds[i]
        if k < stroke_frac:
            # road: open polyline of short steps, heading random walk
            n = int(rng.integers(12, 61))
            step = rng.uniform(1.5, 7.0)
            heading = rng.uniform(0, 2 * math.pi) + np.cumsum(rng.normal(0.0, 0.25, n))
            x0, y0 = rng.uniform(0, size, 2)
            pts = np.empty((n, 2))
            pts[:, 0] = x0 + np.cumsum(np.cos(heading) * step)
            pts[:, 1] = y0 + np.cumsum(np.sin(heading) * step)
            width = math.exp(rng.uniform(math.log(0.5), math.log(3.0)))
            s.stroke(Stroke(width), ident, col, None, _polyline(pts, False))
        elif k < 1.0 - blob_frac:
            # building / park: closed polygon with many short edges
            n = int(rng.integers(8, 77))
            r = rng.uniform(3.0, 28.0)
            cx, cy = rng.uniform(0, size, 2)
            ang = np.sort(rng.uniform(0, 2 * math.pi, n))
            rr = r * rng.uniform(0.7, 1.0, n)
            pts = np.stack([cx + rr * np.cos(ang), cy + rr * np.sin(ang)], axis=1)
            s.fill(Fill.NonZero, ident, col, None, _polyline(pts, True))
        else:
            # curved blob: closed cubic loop
            n = int(rng.integers(4, 13))
            r = rng.uniform(10.0, 60.0)
            cx, cy = rng.uniform(0, size, 2)
            ang = np.linspace(0, 2 * math.pi, n, endpoint=False) + rng.uniform(0, 1)
            rr = r * rng.uniform(0.7, 1.0, n)
            px, py = cx + rr * np.cos(ang), cy + rr * np.sin(ang)
            verbs = [MOVE_TO]
            coords = [px[0], py[0]]
            for j in range(n):
                a, b = j, (j + 1) % n
                t = 0.55 * r * (2 * math.pi / n) / 1.5
                c1 = (px[a] - t * math.sin(ang[a]), py[a] + t * math.cos(ang[a]))
                c2 = (px[b] + t * math.sin(ang[b]), py[b] - t * math.cos(ang[b]))
                verbs.append(CURVE_TO)
                coords.extend([c1[0], c1[1], c2[0], c2[1], px[b], py[b]])
            verbs.append(CLOSE_PATH)
            s.fill(Fill.NonZero, ident, col, None, BezPath.from_arrays(verbs, coords))
    return s


def paris_like_scene_d2(seed=0x5EED0001, n_paths=30000, size=1600.0):
    """C3 exactly as SURVEY.md 8d d2 restates it ("paris-30k-like", SYNTHETIC): 30 000 paths on a 1600x1600 canvas,
    70 % stroked open polylines (width log-uniform 0.5-4 px, 8-60 vertices, step length 4-40 px drawn once per
    polyline, heading random walk sigma = 0.4 rad), 25 % filled closed polygons (6-40 vertices, radius 5-60 px,
    NonZero), 5 % filled cubic blobs; 16 opaque palette colours; kurbo's default stroke style (round joins and caps,
    miter limit 4: an assumption, kurbo is not in the reference tree).  The survey names PCG32; the generator is
    numpy's PCG64 with the survey's seed (0x5EED0001 + GPU rank for C5).  Unlike paris_like_scene this one does NOT
    fit the reference's fixed pools (config.rs:401-408: 2^21 lines / crossings / segments): it needs
    D2_CAPACITIES (vello_hip_capacities), which is what robust dynamic memory (8f f4) exists for."""
    rng = np.random.Generator(np.random.PCG64(seed))
    s = Scene()
    ident = Affine.IDENTITY
    kinds = rng.random(n_paths)
    for i in range(n_paths):
        col = PALETTE[int(rng.integers(0, len(PALETTE)))]
        k = kinds[i]
        if k < 0.70:
            n = int(rng.integers(8, 61))
            step = rng.uniform(4.0, 40.0)
            heading = rng.uniform(0, 2 * math.pi) + np.cumsum(rng.normal(0.0, 0.4, n))
            x0, y0 = rng.uniform(0, size, 2)
            pts = np.empty((n, 2))
            pts[:, 0] = x0 + np.cumsum(np.cos(heading) * step)
            pts[:, 1] = y0 + np.cumsum(np.sin(heading) * step)
            width = math.exp(rng.uniform(math.log(0.5), math.log(4.0)))
            s.stroke(Stroke(width), ident, col, None, _polyline(pts, False))
        elif k < 0.95:
            n = int(rng.integers(6, 41))
            r = rng.uniform(5.0, 60.0)
            cx, cy = rng.uniform(0, size, 2)
            ang = np.sort(rng.uniform(0, 2 * math.pi, n))
            rr = r * rng.uniform(0.7, 1.0, n)
            pts = np.stack([cx + rr * np.cos(ang), cy + rr * np.sin(ang)], axis=1)
            s.fill(Fill.NonZero, ident, col, None, _polyline(pts, True))
        else:
            n = int(rng.integers(4, 13))
            r = rng.uniform(10.0, 60.0)
            cx, cy = rng.uniform(0, size, 2)
            ang = np.linspace(0, 2 * math.pi, n, endpoint=False) + rng.uniform(0, 1)
            rr = r * rng.uniform(0.7, 1.0, n)
            px, py = cx + rr * np.cos(ang), cy + rr * np.sin(ang)
            verbs = [MOVE_TO]
            coords = [px[0], py[0]]
            for j in range(n):
                a, b = j, (j + 1) % n
                t = 0.55 * r * (2 * math.pi / n) / 1.5
                c1 = (px[a] - t * math.sin(ang[a]), py[a] + t * math.cos(ang[a]))
                c2 = (px[b] + t * math.sin(ang[b]), py[b] - t * math.cos(ang[b]))
                verbs.append(CURVE_TO)
                coords.extend([c1[0], c1[1], c2[0], c2[1], px[b], py[b]])
            verbs.append(CLOSE_PATH)
            s.fill(Fill.NonZero, ident, col, None, BezPath.from_arrays(verbs, coords))
    return s


_MMARK_COLORS = [Color.from_rgb8(*c) for c in [(0x10, 0x10, 0x10), (0x80, 0x80, 0x80), (0xc0, 0xc0, 0xc0), (0x10, 0x10, 0x10),
                                               (0x80, 0x80, 0x80), (0xc0, 0xc0, 0xc0), (0xe0, 0x10, 0x40)]]
_OFFSETS = [(-4, 0), (2, 0), (1, -2), (1, 2)]


def mmark_scene(seed=0x5EED0002, n=50000, size=2048.0):
    """C4: port of examples/scenes/src/mmark.rs:43-202 with a seeded RNG (upstream uses rand::rng()),
    coordinates scaled by size/1600, text label omitted (mmark.rs:108-116)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    GRID_W, GRID_H, W, H = 80, 40, 1600.0, 900.0
    k = size / 1600.0

    def rand_point(last):
        ox, oy = _OFFSETS[int(rng.integers(0, 4))]
        x = last[0] + ox
        if not (0 <= x <= GRID_W):
            x -= ox * 2
        y = last[1] + oy
        if not (0 <= y <= GRID_H):
            y -= oy * 2
        return (x, y)

    def coord(g):
        return ((g[0] + 0.5) * (W / (GRID_W + 1)) * k, (100.0 + (g[1] + 0.5) * (H / (GRID_H + 1))) * k)

    s = Scene()
    last = (GRID_W // 2, GRID_H // 2)
    verbs, coords = [], []
    for i in range(n):
        seg_type = int(rng.integers(0, 4))
        nxt = rand_point(last)
        if not verbs:
            verbs.append(MOVE_TO)
            coords.extend(coord(last))
        if seg_type < 2:
            verbs.append(LINE_TO); coords.extend(coord(nxt)); gp = nxt
        elif seg_type < 3:
            p2 = rand_point(nxt)
            verbs.append(QUAD_TO); coords.extend(coord(nxt)); coords.extend(coord(p2)); gp = p2
        else:
            p2 = rand_point(nxt); p3 = rand_point(nxt)
            verbs.append(CURVE_TO); coords.extend(coord(nxt)); coords.extend(coord(p2)); coords.extend(coord(p3)); gp = p3
        color = _MMARK_COLORS[int(rng.integers(0, len(_MMARK_COLORS)))]
        width = (rng.random() ** 5 * 20.0 + 1.0) * k
        is_split = bool(rng.integers(0, 2))
        last = gp
        if is_split or i == n - 1:
            s.stroke(Stroke(width), Affine.IDENTITY, color, None, BezPath.from_arrays(verbs, coords))
            verbs, coords = [], []
    return s


def stroke_styles_scene(size=256.0):
    """Every join x cap combination on open and closed paths (path.rs:847-877 style matrix)."""
    s = Scene()
    y = 20.0
    for join in (Join.Bevel, Join.Miter, Join.Round):
        x = 20.0
        for cap in (Cap.Butt, Cap.Square, Cap.Round):
            p = BezPath()
            p.move_to((x, y)); p.line_to((x + 40, y + 10)); p.line_to((x + 20, y + 40)); p.quad_to((x + 50, y + 60), (x + 60, y + 30))
            p.curve_to((x + 70, y), (x + 30, y - 10), (x + 65, y + 55))
            st = Stroke(6.0, join=join, miter_limit=4.0, start_cap=cap, end_cap=cap)
            s.stroke(st, Affine.IDENTITY, PALETTE[(int(join) * 3 + int(cap)) % 16], None, p)
            q = BezPath()
            q.move_to((x + 5, y + 45)); q.line_to((x + 30, y + 50)); q.line_to((x + 15, y + 70)); q.close_path()
            s.stroke(Stroke(3.0, join=join, start_cap=cap, end_cap=cap), Affine.rotate(0.05), PALETTE[(int(cap) + 5) % 16], None, q)
            x += 78.0
        y += 78.0
    return s


def clip_blend_scene(size=256.0, depth=6):
    """Nested clip layers and blend layers (scene.rs:105-253): exercises clip_reduce/clip_leaf,
    BEGIN/END_CLIP in coarse and the blend stack (+ spill past depth 4) in fine."""
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(40, 60, 200), None, Rect(0, 0, size, size))
    for d in range(depth):
        inset = 10.0 + 14.0 * d
        if d % 2 == 0:
            s.push_clip_layer(Fill.NonZero, Affine.IDENTITY, Circle((size / 2, size / 2), size / 2 - inset))
        else:
            mix = [Mix.Multiply, Mix.Screen, Mix.Overlay, Mix.Difference, Mix.Hue, Mix.SoftLight][d % 6]
            s.push_layer(Fill.EvenOdd if d == 3 else Fill.NonZero, BlendMode(mix, Compose.SrcOver), 0.8, Affine.rotate(0.1 * d),
                         Rect(inset, inset, size - inset, size - inset))
        s.fill(Fill.NonZero, Affine.IDENTITY, PALETTE[d % 16].with_alpha(0.7), None, Rect(inset, size / 2 - 20, size - inset, size / 2 + 20 + 4 * d))
        s.fill(Fill.EvenOdd, Affine.IDENTITY, PALETTE[(d + 7) % 16], None, Circle((size / 2 + 10 * d, size / 3), 25.0))
    for d in range(depth):
        s.pop_layer()
    s.push_layer(Fill.NonZero, BlendMode(Mix.Normal, Compose.Xor), 0.5, Affine.IDENTITY, Rect(0, 0, size / 2, size / 2))
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(255, 255, 0), None, Circle((size / 4, size / 4), size / 5))
    s.pop_layer()
    s.push_luminance_mask_layer(Fill.NonZero, 1.0, Affine.IDENTITY, Rect(size / 2, size / 2, size, size))
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(200, 200, 200), None, Circle((3 * size / 4, 3 * size / 4), size / 6))
    s.pop_layer()
    return s


def random_test_scene(seed, n_paths=200, size=512.0, strokes=True, clips=False):
    """Mixed random scene for differential tests: fills (both rules), strokes (all styles), curves, transforms."""
    rng = np.random.Generator(np.random.PCG64(seed))
    s = Scene()
    open_layers = 0
    for i in range(n_paths):
        col = PALETTE[int(rng.integers(0, 16))].with_alpha(float(rng.choice([1.0, 1.0, 0.6])))
        n = int(rng.integers(2, 12))
        cx, cy = rng.uniform(-20, size + 20, 2)
        r = rng.uniform(2.0, size / 4)
        p = BezPath()
        p.move_to((cx + r, cy))
        for j in range(n):
            kind = int(rng.integers(0, 3))
            a = rng.uniform(0, 2 * math.pi, 3)
            rr = r * rng.uniform(0.3, 1.0, 3)
            pts = [(cx + rr[t] * math.cos(a[t]), cy + rr[t] * math.sin(a[t])) for t in range(3)]
            if kind == 0:
                p.line_to(pts[0])
            elif kind == 1:
                p.quad_to(pts[0], pts[1])
            else:
                p.curve_to(pts[0], pts[1], pts[2])
        if rng.random() < 0.5:
            p.close_path()
        aff = Affine.IDENTITY if rng.random() < 0.6 else Affine.translate(size / 2, size / 2) * Affine.rotate(rng.uniform(0, 6.28)) * Affine.scale(rng.uniform(0.5, 1.5)) * Affine.translate(-size / 2, -size / 2)
        if clips and rng.random() < 0.08 and open_layers < 6:
            s.push_clip_layer(Fill.NonZero, aff, p)
            open_layers += 1
            continue
        if clips and open_layers and rng.random() < 0.08:
            s.pop_layer()
            open_layers -= 1
        if strokes and rng.random() < 0.4:
            st = Stroke(rng.uniform(0.3, 12.0), join=Join(int(rng.integers(0, 3))), miter_limit=rng.uniform(1.0, 8.0),
                        start_cap=Cap(int(rng.integers(0, 3))), end_cap=Cap(int(rng.integers(0, 3))))
            s.stroke(st, aff, col, None, p)
        else:
            s.fill(Fill(int(rng.integers(0, 2))), aff, col, None, p)
    for _ in range(open_layers):
        s.pop_layer()
    return s


def _test_image(w, h, seed, premultiplied=False):
    """Deterministic RGBA8 test pattern with varying alpha (checker + gradient + noise)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), dtype=np.uint8)
    img[..., 0] = (x * 255 // max(w - 1, 1))
    img[..., 1] = (y * 255 // max(h - 1, 1))
    img[..., 2] = (((x // 4) + (y // 4)) % 2) * 200 + rng.integers(0, 56, (h, w))
    img[..., 3] = np.clip(64 + 3 * ((x + y) % 64), 0, 255)
    if premultiplied:
        a = img[..., 3:4].astype(np.uint16)
        img[..., :3] = (img[..., :3].astype(np.uint16) * a // 255).astype(np.uint8)
    return img


def brushes_scene(size=256.0):
    """Gradient, image and blurred-rounded-rect brushes: every gradient kind x extend mode, brush
    transforms, ramps with alpha in both interpolation spaces, images at the three qualities with all extend modes,
    BGRA / premultiplied sources, and blurred rounded rects under rotation."""
    from vello_amd import (Gradient, Extend, InterpolationAlphaSpace, ImageData, ImageBrush, ImageFormat, ImageAlphaType,
                           ImageQuality, RoundedRect)
    s = Scene()
    cs = [Color.from_rgb8(255, 0, 0), Color.from_rgb8(0, 255, 0), Color.from_rgb8(0, 0, 255)]
    ca = [Color.from_rgba8(255, 200, 0, 255), Color.from_rgba8(0, 100, 255, 40), Color.from_rgba8(255, 255, 255, 200)]
    q = size / 4
    # row 0: linear gradients, three extend modes + a brush transform
    for i, ext in enumerate([Extend.Pad, Extend.Repeat, Extend.Reflect]):
        g = Gradient.new_linear((i * q + 10, 10), (i * q + 30, 30)).with_stops(cs).with_extend(ext)
        s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(i * q + 2, 2, (i + 1) * q - 2, q - 2))
    g = Gradient.new_linear((0, 0), (20, 0)).with_stops([(0.0, ca[0]), (0.3, ca[1]), (1.0, ca[2])]).with_extend(Extend.Reflect)
    s.fill(Fill.NonZero, Affine.translate(3 * q, 0), g, Affine.rotate(0.6), Circle((q / 2, q / 2), q / 2 - 3))
    # row 1: radial gradients: circular, two-point (cone), focal-on-circle, strip (equal radii)
    g = Gradient.new_radial((q / 2, 1.5 * q), q / 2.5).with_stops(cs).with_extend(Extend.Repeat)
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(2, q + 2, q - 2, 2 * q - 2))
    g = Gradient.new_two_point_radial((1.3 * q, 1.4 * q), 4.0, (1.6 * q, 1.6 * q), q / 3).with_stops(ca).with_extend(Extend.Pad)
    g = g.with_interpolation_alpha_space(InterpolationAlphaSpace.Unpremultiplied)
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(q + 2, q + 2, 2 * q - 2, 2 * q - 2))
    g = Gradient.new_two_point_radial((2.5 * q - 10, 1.5 * q), 0.0, (2.5 * q, 1.5 * q), 10.0).with_stops(cs).with_extend(Extend.Reflect)
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(2 * q + 2, q + 2, 3 * q - 2, 2 * q - 2))
    g = Gradient.new_two_point_radial((3.3 * q, 1.3 * q), 12.0, (3.7 * q, 1.7 * q), 12.0).with_stops(cs).with_extend(Extend.Pad)
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Rect(3 * q + 2, q + 2, 4 * q - 2, 2 * q - 2))
    # row 2: sweep gradients + a stroked gradient; one- and zero-stop gradients fall back to colours
    g = Gradient.new_sweep((q / 2, 2.5 * q), 0.0, 2 * math.pi).with_stops(cs + [cs[0]])
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, Circle((q / 2, 2.5 * q), q / 2 - 3))
    g = Gradient.new_sweep((1.5 * q, 2.5 * q), 0.5, 2.5).with_stops(ca).with_extend(Extend.Repeat)
    s.fill(Fill.EvenOdd, Affine.IDENTITY, g, None, Rect(q + 2, 2 * q + 2, 2 * q - 2, 3 * q - 2))
    g = Gradient.new_linear((2 * q, 2 * q), (3 * q, 3 * q)).with_stops(cs)
    s.stroke(Stroke(6.0), Affine.IDENTITY, g, None, Circle((2.5 * q, 2.5 * q), q / 2 - 8))
    s.fill(Fill.NonZero, Affine.IDENTITY, Gradient.new_linear((0, 0), (1, 1)).with_stops([cs[1]]), None, Rect(3 * q + 2, 2 * q + 2, 3.5 * q, 3 * q - 2))
    s.fill(Fill.NonZero, Affine.IDENTITY, Gradient.new_linear((0, 0), (1, 1)).with_stops([]), None, Rect(3.5 * q, 2 * q + 2, 4 * q - 2, 3 * q - 2))
    # row 3: images (three qualities, extend modes, BGRA, premultiplied, alpha) and blurred rounded rects
    im_a = ImageData(_test_image(24, 16, 1))
    im_b = ImageData(_test_image(13, 9, 2), ImageFormat.Bgra8)
    im_c = ImageData(_test_image(32, 32, 3, premultiplied=True), ImageFormat.Rgba8, ImageAlphaType.AlphaPremultiplied)
    s.draw_image(ImageBrush(im_a, quality=ImageQuality.Low), Affine.translate(4, 3 * q + 4) * Affine.scale(2.0))
    s.fill(Fill.NonZero, Affine.IDENTITY, ImageBrush(im_b, Extend.Repeat, Extend.Reflect, ImageQuality.Medium, 0.8),
           Affine.translate(q, 3 * q) * Affine.rotate(0.3) * Affine.scale(1.7), Rect(q + 2, 3 * q + 2, 2 * q - 2, 4 * q - 2))
    s.fill(Fill.NonZero, Affine.IDENTITY, ImageBrush(im_c, Extend.Reflect, Extend.Repeat, ImageQuality.High),
           Affine.translate(2 * q + 5, 3 * q + 5) * Affine.scale_non_uniform(0.7, 1.3), Circle((2.5 * q, 3.5 * q), q / 2 - 2))
    s.draw_image(ImageBrush(im_a, quality=ImageQuality.High, alpha=0.5), Affine.translate(2 * q + 8, 3 * q + 30) * Affine.rotate(-0.2))
    s.draw_blurred_rounded_rect(Affine.IDENTITY, (3 * q + 12, 3 * q + 12, 4 * q - 12, 4 * q - 20), Color.from_rgba8(20, 20, 20, 220), 6.0, 3.0)
    s.draw_blurred_rounded_rect(Affine.rotate(0.2), (3 * q + 40, 2 * q - 20, 4 * q + 10, 2 * q + 10), Color.from_rgb8(200, 30, 90), 2.0, 1.2)
    s.draw_blurred_rounded_rect_in(Circle((q / 2, q / 2), q / 3), Affine.IDENTITY, (4, 4, q - 4, q - 4), Color.from_rgba8(0, 0, 0, 160), 0.0, 5.0)
    return s


def heavy_strokes_scene():
    """169 path tags (ONE flatten workgroup) that expand to 3728 lines: wide round strokes of large circles.  Exceeds the
    3072-line LDS staging area of k_flatten, so the tail pieces take the direct-to-soup route."""
    s = Scene()
    for i in range(24):
        s.stroke(Stroke(40.0 + i), Affine.IDENTITY, Color.from_rgba8(20 * i % 255, 100, 200, 200), None,
                 Circle((512 + 3 * i, 512 - 2 * i), 100 + 15 * i))
    return s
