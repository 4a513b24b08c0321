"""Scenes restated from the reference's own test-scene catalogue (examples/scenes/src/test_scenes.rs): the inputs the
upstream snapshot tests render.  The geometry tables are the reference's test vectors; everything is rebuilt on this
repo's Scene API.  Used for oracle-relative parity (the upstream snapshots themselves are Git-LFS stubs)."""
import numpy as np

from vello_amd import Affine, BezPath, Cap, Color, Fill, Join, Rect, Scene, Stroke

# test_scenes.rs:549-576 (tricky_strokes): cusps, near-cusps, 180-degree turns, degenerate and flat cubics
TRICKY_CUBICS = [
    [(122., 737.), (348., 553.), (403., 761.), (400., 760.)],
    [(244., 520.), (244., 518.), (1141., 634.), (394., 688.)],
    [(550., 194.), (138., 130.), (1035., 246.), (288., 300.)],
    [(226., 733.), (556., 779.), (-43., 471.), (348., 683.)],
    [(268., 204.), (492., 304.), (352., 23.), (433., 412.)],
    [(172., 480.), (396., 580.), (256., 299.), (338., 677.)],
    [(731., 340.), (318., 252.), (1026., -64.), (367., 265.)],
    [(475., 708.), (62., 620.), (770., 304.), (220., 659.)],
    [(0., 0.), (128., 128.), (128., 0.), (0., 128.)],
    [(0., 0.01), (128., 127.999), (128., 0.01), (0., 127.99)],
    [(0., -0.01), (128., 128.001), (128., -0.01), (0., 128.001)],
    [(0., 0.), (0., -10.), (0., -10.), (0., 10.)],
    [(10., 0.), (0., 0.), (20., 0.), (10., 0.)],
    [(39., -39.), (40., -40.), (40., -40.), (0., 0.)],
    [(40., 40.), (0., 0.), (200., 200.), (0., 0.)],
    [(0., 0.), (1e-2, 0.), (-1e-2, 0.), (0., 0.)],
    [(400.75, 100.05), (400.75, 100.05), (100.05, 300.95), (100.05, 300.95)],
    [(0.5, 0.), (0., 0.), (20., 0.), (10., 0.)],
    [(10., 0.), (0., 0.), (10., 0.), (10., 0.)],
]
# test_scenes.rs:579-582, :621-638: flat quads with cusps (the 1.5-weight conic lowered to quads)
FLAT_QUAD = [[(2., 1.), (1., 1.)]]
BIGGER_FLAT_CONIC_AS_QUADS = [
    [(8.979845, 1.0), (15.795975, 1.0)], [(22.612104, 1.0), (28.363287, 1.0)], [(34.114471, 1.0), (38.884045, 1.0)],
    [(43.653618, 1.0), (47.510696, 1.0)], [(51.367767, 1.0), (54.368233, 1.0)], [(57.368698, 1.0), (59.556030, 1.0)],
    [(61.743366, 1.0), (63.149269, 1.0)], [(64.555168, 1.0), (65.200005, 1.0)], [(65.844841, 1.0), (65.737961, 1.0)],
    [(65.631073, 1.0), (64.770912, 1.0)], [(63.910763, 1.0), (62.284878, 1.0)], [(60.658997, 1.0), (58.243816, 1.0)],
    [(55.828640, 1.0), (52.589172, 1.0)], [(49.349705, 1.0), (45.239006, 1.0)], [(41.128315, 1.0), (36.086826, 1.0)],
    [(31.045338, 1.0), (25.000000, 1.0)],
]
_COLORS = [(140, 181, 236), (246, 236, 202), (201, 147, 206), (150, 195, 160)]


def _cubic_bounds(pts, n=512):
    t = np.linspace(0.0, 1.0, n)[:, None]
    p = [np.array(q, dtype=np.float64) for q in pts]
    c = ((1 - t) ** 3) * p[0] + 3 * ((1 - t) ** 2) * t * p[1] + 3 * (1 - t) * t * t * p[2] + (t ** 3) * p[3]
    return c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()


def _map_rect_to_rect(src, dst):
    sw, sh = max(src[2] - src[0], 1e-9), max(src[3] - src[1], 1e-9)
    dw, dh = dst[2] - dst[0], dst[3] - dst[1]
    sx, sy = dw / sw, dh / sh
    scale = min(sx, sy)
    tx, ty = dst[0] - src[0] * scale, dst[1] - src[1] * scale
    if sx > sy:
        tx += 0.5 * (dw - sw * scale)
    else:
        ty += 0.5 * (dh - sh * scale)
    return Affine((scale, 0.0, 0.0, scale, tx, ty)), scale


def tricky_strokes_scene(join=Join.Miter, cap=Cap.Butt):
    """test_scenes.rs:513-697; returns (scene, width, height).  Cell bounds come from a sampled bounding box instead of
    kurbo's exact CubicBez::bounding_box (kurbo is not in the tree): placement differs by a fraction of a pixel."""
    cell, sw, cols = 200.0, 30.0, 5
    s = Scene()
    idx = 0
    for i, cub in enumerate(TRICKY_CUBICS):
        x, y = (i % cols) * cell, (i // cols) * cell
        b = _cubic_bounds(cub)
        t, sc = _map_rect_to_rect((b[0] - sw, b[1] - sw, b[2] + sw, b[3] + sw), (x, y, x + cell, y + cell))
        p = BezPath()
        p.move_to(cub[0])
        p.curve_to(cub[1], cub[2], cub[3])
        s.stroke(Stroke(sw / sc).with_caps(cap).with_join(join), t, Color.from_rgb8(*_COLORS[i % 4]), None, p)
        idx += 1
    for quads in (FLAT_QUAD, BIGGER_FLAT_CONIC_AS_QUADS):
        p = BezPath()
        p.move_to((1.0, 1.0))
        xs = [1.0]
        for q in quads:
            p.quad_to(q[0], q[1])
            xs += [q[0][0], q[1][0]]
        x, y = (idx % cols) * cell, (idx // cols) * cell
        t, sc = _map_rect_to_rect((min(xs) - sw, 1.0 - sw, max(xs) + sw, 1.0 + sw), (x, y, x + cell, y + cell))
        s.stroke(Stroke(sw / sc).with_caps(cap).with_join(join), t, Color.from_rgb8(*_COLORS[idx % 4]), None, p)
        idx += 1
    n = len(TRICKY_CUBICS) + 2
    return s, int(cell * cols), int(cell * (1 + n // cols))


def fill_types_scene():
    """test_scenes.rs:699-770 without the text labels: self-intersecting star and overlapping arcs under both fill rules,
    then the same with rotated translucent copies.  1400 x 700."""
    s = Scene()
    rect = Rect(0.0, 0.0, 500.0, 500.0)
    star = BezPath()
    star.move_to((250., 0.)); star.line_to((105., 450.)); star.line_to((490., 175.)); star.line_to((10., 175.))
    star.line_to((395., 450.)); star.close_path()
    arcs = BezPath()
    arcs.move_to((0., 480.)); arcs.curve_to((500., 480.), (500., -10.), (0., -10.)); arcs.close_path()
    arcs.move_to((500., -10.)); arcs.curve_to((0., -10.), (0., 480.), (500., 480.)); arcs.close_path()
    gray, yellow = Color.from_rgb8(128, 128, 128), Color.from_rgb8(255, 255, 0)
    rules = [(Fill.NonZero, star), (Fill.EvenOdd, star), (Fill.NonZero, arcs), (Fill.EvenOdd, arcs)]
    scale = Affine.scale(0.6)
    t0 = Affine.translate(10., 25.)
    for blends in (False, True):
        tb = Affine.translate(700., 0.) * t0 if blends else t0
        for i, (rule, path) in enumerate(rules):
            t = Affine.translate((i % 2) * 306., (i // 2) * 340.) * tb
            t = Affine.translate(0., 5.) * t * scale
            s.fill(Fill.NonZero, t, gray, None, rect)
            s.fill(rule, Affine.translate(0., 10.) * t, yellow, None, path)
            if blends:
                s.fill(rule, Affine.translate(0., 10.) * t * Affine.rotate(0.06), Color(0., 1., 0.7, 0.6), None, path)
                s.fill(rule, Affine.translate(0., 10.) * t * Affine.rotate(-0.06), Color(0.9, 0.7, 0.5, 0.6), None, path)
    return s, 1400, 700


def robust_paths_scene():
    """test_scenes.rs:1610-1691: axis-aligned and tile-boundary-aligned polygons (edges exactly on multiples of 16),
    the cases the ROBUST_EPSILON / ONE_MINUS_ULP handling exists for.  600 x 160."""
    p = BezPath()
    polys = [
        [(16, 16), (32, 16), (32, 32), (16, 32)],
        [(48, 18), (64, 23), (64, 33), (48, 38)],
        [(80, 18), (82, 16), (94, 16), (96, 18), (96, 30), (94, 32), (82, 32), (80, 30)],
        [(112, 16), (128, 16), (128, 32)],
        [(144, 16), (160, 32), (144, 32)],
        [(168, 8), (184, 8), (184, 24)],
        [(200, 8), (216, 24), (200, 24)],
        [(241, 17.5), (255, 17.5), (255, 19.5), (241, 19.5)],
        [(241, 22.5), (256, 22.5), (256, 24.5), (241, 24.5)],
    ]
    for poly in polys:
        p.move_to(poly[0])
        for q in poly[1:]:
            p.line_to(q)
        p.close_path()
    s = Scene()
    yellow, lime = Color.from_rgb8(255, 255, 0), Color.from_rgb8(0, 255, 0)
    s.fill(Fill.NonZero, Affine.IDENTITY, yellow, None, p)
    s.fill(Fill.EvenOdd, Affine.translate(300.0, 0.0), lime, None, p)
    p.move_to((8.0, 4.0)); p.line_to((8.0, 40.0)); p.line_to((260.0, 40.0)); p.line_to((260.0, 4.0)); p.close_path()
    s.fill(Fill.NonZero, Affine.translate(0.0, 100.0), yellow, None, p)
    s.fill(Fill.EvenOdd, Affine.translate(300.0, 100.0), lime, None, p)
    return s, 600, 160


def gradient_extend_scene():
    """test_scenes.rs:978-1043 without the labels: linear / two-point radial / sweep gradients under the three extend
    modes.  1200 x 1200."""
    from vello_amd import Extend, Gradient
    s = Scene()
    colors = [Color.from_rgb8(255, 0, 0), Color.from_rgb8(0, 255, 0), Color.from_rgb8(0, 0, 255)]
    w = h = 300.0
    for x, ext in enumerate([Extend.Pad, Extend.Repeat, Extend.Reflect]):
        for y in range(3):
            if y == 0:
                g = Gradient.new_linear((w * 0.35, h * 0.5), (w * 0.65, h * 0.5))
            elif y == 1:
                radius = float(np.float32(w * 0.25))
                g = Gradient.new_two_point_radial((w * 0.5, h * 0.5), radius * 0.25, (w * 0.5, h * 0.5), radius)
            else:
                g = Gradient.new_sweep((w * 0.5, h * 0.5), float(np.radians(np.float32(30.0))), float(np.radians(np.float32(150.0))))
            g = g.with_stops(colors).with_extend(ext)
            s.fill(Fill.NonZero, Affine.translate(x * 350.0 + 50.0, y * 350.0 + 100.0), g, None, Rect(0.0, 0.0, w, h))
    return s, 1200, 1200


def blend_grid_scene():
    """test_scenes.rs:1213-1239 + render_blend_square (:1398-1436): the 16 mix modes, each over gradient backdrops with
    gradient-filled squashed ellipses inside nested blend layers.  900 x 900."""
    from vello_amd import Circle, Gradient, Mix, BlendMode, Compose
    import math
    modes = [Mix.Normal, Mix.Multiply, Mix.Darken, Mix.Screen, Mix.Lighten, Mix.Overlay, Mix.ColorDodge, Mix.ColorBurn,
             Mix.HardLight, Mix.SoftLight, Mix.Difference, Mix.Exclusion, Mix.Hue, Mix.Saturation, Mix.Color, Mix.Luminosity]
    black, white = Color.from_rgb8(0, 0, 0), Color.from_rgb8(255, 255, 255)
    rect = Rect(0.0, 0.0, 200.0, 200.0)

    def blend_square(mix):
        f = Scene()
        t = Affine.IDENTITY
        f.fill(Fill.NonZero, t, Gradient.new_linear((0.0, 0.0), (200.0, 0.0)).with_stops([black, white]), None, rect)
        for x, y, c in [(150., 0., (255, 240, 64)), (175., 100., (255, 96, 240)), (125., 200., (64, 192, 255))]:
            col = Color.from_rgb8(*c)
            f.fill(Fill.NonZero, t, Gradient.new_radial((x, y), 100.0).with_stops([col, col.with_alpha(0.0)]), None, rect)
        f.push_layer(Fill.NonZero, BlendMode(Mix.Normal, Compose.SrcOver), 1.0, t, rect)
        for i, c in enumerate([(255, 0, 0), (0, 255, 0), (0, 0, 255)]):
            lin = Gradient.new_linear((0.0, 0.0), (0.0, 200.0)).with_stops([white, Color.from_rgb8(*c)])
            f.push_layer(Fill.NonZero, BlendMode(mix, Compose.SrcOver), 1.0, t, rect)
            a = (t * Affine.translate(100., 100.) * Affine.rotate(math.pi / 3.0 * (i * 2 + 1)) * Affine.scale_non_uniform(1.0, 0.357)
                 * Affine.translate(-100., -100.))
            f.fill(Fill.NonZero, a, lin, None, Circle((100., 100.), 90.))
            f.pop_layer()
        f.pop_layer()
        return f

    s = Scene()
    for ix, mix in enumerate(modes):
        s.append(blend_square(mix), Affine.translate((ix % 4) * 225., (ix // 4) * 225.))
    return s, 900, 900


def deep_blend_scene(complexity=7):
    """test_scenes.rs:1241-1276: nested 0.9-alpha layers, deeper than the 4-entry register blend stack.  1000 x 1000."""
    from vello_amd import Mix, BlendMode, Compose
    s = Scene()
    main_rect = Rect(10., 10., 910., 910.)
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(255, 0, 0), None, main_rect)
    options = [(800., (0, 255, 255)), (700., (255, 0, 0)), (600., (240, 248, 255)), (500., (255, 255, 0)), (400., (0, 128, 0)),
               (300., (0, 0, 255)), (200., (255, 165, 0)), (100., (255, 255, 255))]
    depth = 0
    for width, color in options[:min(complexity, len(options) - 1)]:
        s.push_layer(Fill.NonZero, BlendMode(Mix.Normal, Compose.SrcOver), 0.9, Affine.IDENTITY, Rect(10., 10., 10. + width, 10. + width))
        s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(*color), None, main_rect)
        depth += 1
    for _ in range(depth):
        s.pop_layer()
    return s, 1000, 1000


def many_clips_scene(seed=42):
    """test_scenes.rs:1278-1304: 100 triangles, each under three randomly rotated triangular clip layers (600 clips).
    The reference draws its angles from rand's StdRng; here a seeded PCG64 (different values, same structure)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tri = BezPath()
    tri.move_to((-50.0, 0.0)); tri.line_to((25.0, -43.3)); tri.line_to((25.0, 43.3))
    s = Scene()
    for y in range(10):
        for x in range(10):
            tr = Affine.translate(100. * (x + 0.5), 100. * (y + 0.5))
            for _ in range(3):
                s.push_clip_layer(Fill.NonZero, tr * Affine.rotate(float(rng.uniform(0.0, np.pi))), tri)
            rot = Affine.rotate(float(rng.uniform(0.0, np.pi)))
            col = Color(float(rng.random()), float(rng.random()), float(rng.random()), 1.0)
            s.fill(Fill.NonZero, tr * rot, col, None, tri)
            for _ in range(3):
                s.pop_layer()
    return s, 1000, 1000


def blurred_rounded_rect_scene(std_dev=50.0):
    """test_scenes.rs:1988-2031 at time 0 (std_dev = 50): plain, skewed, circle-like and over-rounded blurred rects on
    white.  (The emulated box shadow of :2033-2050 needs kurbo's reverse_subpaths and is left out.)  1200 x 1200."""
    import math
    s = Scene()
    rect = (-150.0, -120.0, 150.0, 120.0)
    blue, black = Color.from_rgb8(0, 0, 255), Color.from_rgb8(0, 0, 0)
    s.draw_blurred_rounded_rect(Affine.translate(300.0, 300.0), rect, blue, 50.0, std_dev)
    s.draw_blurred_rounded_rect(Affine.translate(900.0, 300.0) * Affine.skew(math.tan(math.radians(20.0)), 0.0), rect, black, 50.0, std_dev)
    s.draw_blurred_rounded_rect(Affine.IDENTITY, (100.0, 800.0, 400.0, 1100.0), black, 150.0, std_dev)
    s.draw_blurred_rounded_rect(Affine.IDENTITY, (600.0, 800.0, 900.0, 900.0), black, 150.0, std_dev)
    return s, 1200, 1200


def _rgba(*c):
    return np.array(c, dtype=np.uint8)


def image_sampling_scene():
    """test_scenes.rs:141-161, :2053-2084: a 2x2 image magnified 200x under scale, 45-degree rotation, non-uniform scale
    and skew (bilinear).  1100 x 1100 on white."""
    import math
    from vello_amd import ImageData
    px = np.zeros((2, 2, 4), dtype=np.uint8)
    px[0, 0], px[0, 1], px[1, 0], px[1, 1] = _rgba(255, 0, 0, 255), _rgba(0, 0, 255, 255), _rgba(0, 255, 255, 255), _rgba(255, 0, 255, 255)
    image = ImageData(px)
    s = Scene()
    s.draw_image(image, Affine.scale(200.0).then_translate(100.0, 100.0))
    s.draw_image(image, Affine.translate(-1.0, -1.0).then_rotate(math.pi / 4.0).then_translate(1.0, 1.0)
                 .then_scale(200.0 * math.sqrt(0.5)).then_translate(100.0, 600.0))
    s.draw_image(image, Affine.scale_non_uniform(100.0, 200.0).then_translate(600.0, 100.0))
    s.draw_image(image, Affine.skew(0.1, 0.25).then_scale(200.0).then_translate(600.0, 600.0))
    return s, 1100, 1100


def image_sampling_bicubic_scene():
    """test_scenes.rs:163-193, :2086-2113: a 16x16 checker with red/blue/lime features at Low / Medium / High quality
    under a rotation and a skew.  1400 x 900 on white."""
    import math
    from vello_amd import ImageData, ImageBrush, ImageQuality
    px = np.zeros((16, 16, 4), dtype=np.uint8)
    for y in range(16):
        for x in range(16):
            c = (0, 0, 0) if ((x // 2) + (y // 2)) % 2 == 0 else (255, 255, 255)
            if x == 8 or y == 8:
                c = (255, 0, 0)
            if x == y or x + y == 15:
                c = (0, 0, 255)
            if (x == 2 and y == 13) or (x == 13 and y == 2):
                c = (0, 255, 0)
            px[y, x] = _rgba(*c, 255)
    image = ImageData(px)
    brushes = [ImageBrush(image, quality=q) for q in (ImageQuality.Low, ImageQuality.Medium, ImageQuality.High)]
    transforms = [
        Affine.translate(-8.0, -8.0).then_rotate(math.pi / 5.0).then_scale_non_uniform(18.0, 14.0).then_translate(250.0, 270.0),
        Affine.translate(250.0, 670.0) * Affine.scale_non_uniform(20.0, 10.0) * Affine.skew(0.35, -0.15) * Affine.translate(-8.0, -8.0),
    ]
    s = Scene()
    for t in transforms:
        for k, b in enumerate(brushes):
            s.draw_image(b, t.then_translate(420.0 * k, 0.0))
    return s, 1400, 900


def funky_paths_scene():
    """test_scenes.rs:293-333: path-encoder edge cases -- segments after a ClosePath without a MoveTo, a path of
    MoveTos only, an empty path -- filled and stroked.  600 x 600."""
    def missing_movetos():
        p = BezPath()
        p.move_to((0., 0.)); p.line_to((100., 100.)); p.line_to((100., 200.)); p.close_path()
        p.line_to((0., 400.)); p.line_to((100., 400.))
        return p
    only_movetos = BezPath()
    only_movetos.move_to((0., 0.)); only_movetos.move_to((100., 100.))
    s = Scene()
    blue, aqua = Color.from_rgb8(0, 0, 255), Color.from_rgb8(0, 255, 255)
    s.fill(Fill.NonZero, Affine.translate(100., 100.), blue, None, missing_movetos())
    s.fill(Fill.NonZero, Affine.IDENTITY, blue, None, BezPath())
    s.fill(Fill.NonZero, Affine.IDENTITY, blue, None, only_movetos)
    s.stroke(Stroke(8.0), Affine.translate(100., 100.), aqua, None, missing_movetos())
    return s, 600, 600


def cardioid_scene():
    """test_scenes.rs:1306-1331: 600 chords of a circle (a cardioid envelope) as ONE stroked path of 600 two-point
    subpaths: long lines across hundreds of tiles.  2048 x 1536."""
    import math
    n = 601
    dth = math.pi * 2.0 / n
    cx, cy, r = 1024.0, 768.0, 750.0
    p = BezPath()
    for i in range(1, n):
        a0, a1 = i * dth, ((i * 2) % n) * dth
        p.move_to((cx + math.cos(a0) * r, cy + math.sin(a0) * r))
        p.line_to((cx + math.cos(a1) * r, cy + math.sin(a1) * r))
    s = Scene()
    s.stroke(Stroke(2.0), Affine.IDENTITY, Color.from_rgb8(0, 0, 255), None, p)
    return s, 2048, 1536


def many_draw_objects_scene(n_wide=300, n_high=300):
    """test_scenes.rs:1928-1948: 90 000 little circles (360 000 cubics), one draw object each.  2000 x 1500."""
    from vello_amd import Circle
    s = Scene()
    yellow = Color.from_rgb8(255, 255, 0)
    for j in range(n_high):
        y = (j + 0.5) * (1500.0 / n_high)
        for i in range(n_wide):
            s.fill(Fill.NonZero, Affine.IDENTITY, yellow, None, Circle(((i + 0.5) * (2000.0 / n_wide), y), 3.0))
    return s, 2000, 1500


def ref_stroke_styles_scene(transform=None):
    """test_scenes.rs:335-511 without the labels (fonts) and the dashed column (kurbo::dash is host-side upstream): cap
    combinations, cap x join combinations on a curved path, miter limits and closed paths, each under `transform`
    (identity, Affine.scale_non_uniform(1.2, 0.7) and Affine.skew(1, 0) in the catalogue).  2450 x 1700."""
    from vello_amd import Circle
    transform = transform or Affine.IDENTITY
    colors = [Color.from_rgb8(*c) for c in _COLORS]
    simple = BezPath(); simple.move_to((0., 0.)); simple.line_to((100., 0.))
    join_path = BezPath(); join_path.move_to((0., 0.))
    join_path.curve_to((20., 0.), (42.5, 5.), (50., 25.)); join_path.curve_to((57.5, 5.), (80., 0.), (100., 0.))
    miter = BezPath(); miter.move_to((0., 0.))
    for p in [(90., 16.), (0., 31.), (90., 46.)]:
        miter.line_to(p)
    closed = BezPath()
    closed.move_to((0., 0.)); closed.line_to((90., 21.)); closed.line_to((0., 42.)); closed.close_path()
    closed.move_to((200., 0.)); closed.curve_to((100., 72.), (300., 72.), (200., 0.)); closed.close_path()
    closed.move_to((290., 0.)); closed.curve_to((200., 72.), (400., 72.), (310., 0.)); closed.close_path()
    caps = [Cap.Butt, Cap.Square, Cap.Round]
    joins = [Join.Bevel, Join.Miter, Join.Round]
    s = Scene()
    ci = 0
    t = Affine.translate(60., 40.) * Affine.scale(2.)
    y = 0.
    for start in caps:
        for end in caps:
            s.stroke(Stroke(20., start_cap=start, end_cap=end), Affine.translate(0., y + 30.) * t * transform, colors[ci], None, simple)
            y += 180.
            ci = (ci + 1) % 4
    ci = (ci + 9) % 4  # the dashed column advances the colour cycle too
    t = Affine.translate(550., 0.) * (Affine.translate(450., 0.) * t)
    y = 0.
    for cap in caps:
        for join in joins:
            s.stroke(Stroke(20., join=join, start_cap=cap, end_cap=cap), Affine.translate(0., y + 30.) * t * transform, colors[ci], None,
                     join_path)
            y += 185.
            ci = (ci + 1) % 4
    t = Affine.translate(500., 0.) * t
    y = 0.
    for ml in [4., 6., 0.1, 10.]:
        s.stroke(Stroke(10., join=Join.Miter, miter_limit=ml, start_cap=Cap.Butt, end_cap=Cap.Butt), Affine.translate(0., y + 30.) * t * transform,
                 colors[ci], None, miter)
        y += 180.
        ci = (ci + 1) % 4
    for i, join in enumerate(joins):
        s.stroke(Stroke(10., join=join, miter_limit=5., start_cap=caps[i], end_cap=caps[i]), Affine.translate(0., y + 30.) * t * transform,
                 colors[ci], None, closed)
        y += 180.
        ci = (ci + 1) % 4
    return s, 2450, 1700


def two_point_radial_scene():
    """test_scenes.rs:1045-1211: the COLR radial-gradient cases -- small-to-large, large-to-small, equal radii (strip),
    offset focal circle, and circles touching on the outside (focal on circle) -- each under Pad / Repeat / Reflect, with
    the two circles stroked on top (kurbo::Ellipse with equal radii = a circle).  1300 x 1120."""
    import math
    from vello_amd import Circle, Extend, Gradient
    colors = [Color.from_rgb8(255, 0, 0), Color.from_rgb8(255, 255, 0), Color.from_rgb8(6, 85, 186)]
    s = Scene()

    def make(x0, y0, r0, x1, y1, r1, transform, extend):
        rect = Rect(0.0, 0.0, 400.0, 200.0)
        s.fill(Fill.NonZero, transform, Color.from_rgb8(255, 255, 255), None, rect)
        g = Gradient.new_two_point_radial((x0, y0), np.float32(r0), (x1, y1), np.float32(r1)).with_stops(colors).with_extend(extend)
        s.fill(Fill.NonZero, transform, g, None, rect)
        s.stroke(Stroke(1.0), transform, Color.from_rgb8(0, 0, 0), None, Circle((x0, y0), float(np.float32(r0)) - 1.0))
        s.stroke(Stroke(1.0), transform, Color.from_rgb8(0, 0, 0), None, Circle((x1, y1), float(np.float32(r1)) - 1.0))

    modes = [Extend.Pad, Extend.Repeat, Extend.Reflect]
    for i, mode in enumerate(modes):
        make(140.0, 100.0, 20.0, 280.0, 100.0, 50.0, Affine.translate(i * 420.0 + 20.0, 20.0), mode)
    for i, mode in enumerate(modes):
        make(280.0, 100.0, 50.0, 140.0, 100.0, 20.0, Affine.translate(i * 420.0 + 20.0, 240.0), mode)
    for i, mode in enumerate(modes):
        make(140.0, 100.0, 50.0, 280.0, 100.0, 50.0, Affine.translate(i * 420.0 + 20.0, 460.0), mode)
    for i, mode in enumerate(modes):
        make(140.0, 125.0, 20.0, 190.0, 100.0, 95.0, Affine.translate(i * 420.0 + 20.0, 680.0), mode)
    for i, mode in enumerate(modes):
        x0, y0, r0, x1, y1, r1 = 140.0, 125.0, 20.0, 190.0, 100.0, 96.0
        dx, dy = x0 - x1, y0 - y1
        n = math.hypot(dx, dy)
        make(x1 + dx / n * (r1 - r0), y1 + dy / n * (r1 - r0), r0, x1, y1, r1, Affine.translate(i * 420.0 + 20.0, 900.0), mode)
    return s, 1300, 1120


def conflation_artifacts_scene():
    """test_scenes.rs:1444-1531: shapes whose shared edges conflate under area coverage -- two triangles of opposite
    winding sharing a diagonal, and adjacent rectangles (opposite / same winding, even-odd) at a half-pixel offset.
    300 x 700."""
    N, S = 50.0, 4.0
    scale = Affine.scale(S)
    x, y = N + 0.5, N
    bg, fg = Color.from_rgb8(255, 194, 19), Color.from_rgb8(12, 165, 255)
    s = Scene()
    p = BezPath()
    p.move_to((0., 0.)); p.line_to((N, N)); p.line_to((0., N)); p.line_to((0., 0.))
    p.move_to((0., 0.)); p.line_to((N, N)); p.line_to((N, 0.)); p.line_to((0., 0.))
    s.fill(Fill.NonZero, Affine.translate(x, y) * scale, fg, None, p)
    y += S * N + 10.0
    s.fill(Fill.EvenOdd, Affine.translate(x, y) * scale, bg, None, Rect(0., 0., N, N))
    p = BezPath()
    p.move_to((0., 0.)); p.line_to((0., N)); p.line_to((N * 0.5, N)); p.line_to((N * 0.5, 0.))
    p.move_to((N * 0.5, 0.)); p.line_to((N, 0.)); p.line_to((N, N)); p.line_to((N * 0.5, N))
    s.fill(Fill.EvenOdd, Affine.translate(x, y) * scale, fg, None, p)
    y += S * N + 10.0
    s.fill(Fill.EvenOdd, Affine.translate(x, y) * scale, bg, None, Rect(0., 0., N, N))
    p = BezPath()
    p.move_to((0., 0.)); p.line_to((0., N)); p.line_to((N * 0.5, N)); p.line_to((N * 0.5, 0.))
    p.move_to((N * 0.5, 0.)); p.line_to((N * 0.5, N)); p.line_to((N, N)); p.line_to((N, 0.))
    s.fill(Fill.EvenOdd, Affine.translate(x, y) * scale, fg, None, p)
    return s, 300, 700


_LABYRINTH_ROWS = [
    [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], [0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 1, 1], [0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 1],
    [1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0], [0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1], [1, 0, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0],
    [0, 1, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0], [1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1], [0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1],
    [0, 1, 1, 1, 0, 0, 1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
]
_LABYRINTH_COLS = [
    [1, 1, 1, 1, 0, 1, 1, 1, 1, 1], [0, 0, 1, 0, 0, 0, 1, 1, 1, 0], [0, 1, 1, 0, 1, 1, 1, 0, 0, 1], [1, 1, 0, 0, 0, 0, 1, 0, 1, 0],
    [0, 0, 1, 0, 1, 0, 0, 0, 0, 1], [0, 0, 1, 1, 1, 0, 0, 0, 1, 0], [0, 1, 0, 1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 0, 1, 1, 1, 0, 1, 0],
    [1, 1, 0, 1, 1, 0, 0, 0, 1, 0], [0, 0, 1, 0, 0, 0, 0, 0, 0, 1], [0, 0, 1, 1, 0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 0, 1, 0, 0, 1],
    [1, 1, 1, 1, 1, 1, 0, 1, 1, 1],
]


def labyrinth_scene():
    """test_scenes.rs:1533-1608: ONE non-zero path of 140-odd thin overlapping rectangles (wall segments 0.2 wide on a
    unit grid, scaled by 80 at a half-pixel offset): many sub-paths whose edges coincide and overlap.  1000 x 850."""
    p = BezPath()
    for y, row in enumerate(_LABYRINTH_ROWS):
        for x, flag in enumerate(row):
            if flag:
                p.move_to((x - 0.1, y + 0.1)); p.line_to((x + 1.1, y + 0.1)); p.line_to((x + 1.1, y - 0.1)); p.line_to((x - 0.1, y - 0.1))
    for x, col in enumerate(_LABYRINTH_COLS):
        for y, flag in enumerate(col):
            if flag:
                p.move_to((x - 0.1, y - 0.1)); p.line_to((x - 0.1, y + 1.1)); p.line_to((x + 0.1, y + 1.1)); p.line_to((x + 0.1, y - 0.1))
    s = Scene()
    s.fill(Fill.NonZero, Affine.translate(20.5, 20.5) * Affine.scale(80.0), Color.from_rgb8(0x70, 0x80, 0x80), None, p)
    return s, 1000, 850


def _pentagram(cx, cy, r):
    import math
    pts = [(cx + math.cos(-math.pi / 2 + i * 2 * math.pi / 5) * r, cy + math.sin(-math.pi / 2 + i * 2 * math.pi / 5) * r) for i in range(5)]
    p = BezPath()
    p.move_to(pts[0])
    for i in (2, 4, 1, 3):
        p.line_to(pts[i])
    p.close_path()
    return p


def clip_test_scene():
    """test_scenes.rs:1708-1911 without the clipped text (fonts) and the dashed-stroke clip (kurbo::dash): an even-odd
    clip layer of a self-intersecting star, a clip layer whose style is a STROKE (clip to the stroked outline) over a
    gradient, and the blend layer whose clip rect cuts two of its three fills (the issue #1198 transform values).
    500 x 900."""
    from vello_amd import BlendMode, Compose, Gradient, Mix
    s = Scene()
    demo = Rect(250.0, 20.0, 450.0, 220.0)
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(0, 0, 255), None, demo)
    s.push_clip_layer(Fill.EvenOdd, Affine.IDENTITY, _pentagram(350.0, 120.0, 90.0))
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(255, 0, 0), None, demo)
    s.pop_layer()
    demo = Rect(250.0, 240.0, 450.0, 440.0)
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(112, 128, 144), None, demo)
    s.push_clip_layer(Stroke(18.0, join=Join.Round, start_cap=Cap.Round, end_cap=Cap.Round), Affine.IDENTITY, _pentagram(350.0, 340.0, 85.0))
    g = Gradient.new_linear((250.0, 240.0), (450.0, 440.0)).with_stops([Color.from_rgb8(255, 0, 255), Color.from_rgb8(0, 255, 255)])
    s.fill(Fill.NonZero, Affine.IDENTITY, g, None, demo)
    s.pop_layer()
    scale = 2.0
    s.push_layer(Fill.NonZero, BlendMode(Mix.Normal, Compose.SrcOver), 1.0, Affine([scale, 0.0, 0.0, scale, 27.07470703125, 176.40660533027858]),
                 Rect(0.0, 0.0, 74.4, 339.20000000000005))
    s.fill(Fill.NonZero, Affine([scale, 0.0, 0.0, scale, 27.07470703125, 176.40660533027858]), Color.from_rgb8(0, 0, 255), None,
           Rect(-1000.0, -1000.0, 2000.0, 2000.0))
    s.fill(Fill.NonZero, Affine([scale, 0.0, 0.0, scale, 29.027636718750003, 182.9755506427786]), Color.from_rgb8(0, 255, 0), None,
           Rect(11.0, 13.399999999999999, 59.0, 56.6))
    s.fill(Fill.NonZero, Affine([scale, 0.0, 0.0, scale, 29.027636718750003, scale * 559.3583631427786]), Color.from_rgb8(255, 0, 0), None,
           Rect(12.599999999999998, 12.599999999999998, 57.400000000000006, 57.400000000000006))
    s.pop_layer()
    return s, 500, 900


def luminance_mask_scene():
    """test_scenes.rs:2214-2289 (the MDN mask-type example): red under a luminance-mask layer holding a dark translucent
    square and a light translucent circle, inside a src-over layer over white.  55 x 55."""
    from vello_amd import BlendMode, Circle, Compose, Mix
    s = Scene()
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(255, 255, 255), None, Rect(0., 0., 60., 60.))
    box = Rect(5., 5., 50., 50.)
    s.push_layer(Fill.NonZero, BlendMode(Mix.Normal, Compose.SrcOver), 1.0, Affine.IDENTITY, box)
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(255, 0, 0), None, box)
    s.push_luminance_mask_layer(Fill.NonZero, 1.0, Affine.IDENTITY, box)
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color(0.1, 0.1, 0.1, 0.4), None, box)
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color(0.9, 0.9, 0.9, 0.6), None, Circle((0., 55.), 35.))
    s.pop_layer()
    s.pop_layer()
    return s, 55, 55


def image_luminance_mask_scene(scale=1):
    """test_scenes.rs:2291-2349: a 640 x 480 image under a luminance-mask layer, over beige and aquamarine rectangles, inside a
    src-over layer; 700 x 500.  The reference's image is a JPEG asset (assets/splash-flower.jpg: no decoder here, and what the scene
    tests is the mask, not the flower): a SYNTHETIC opaque image of the same size stands in -- a radial luminance ramp crossed by
    hue stripes, so that the mask takes every value.  scale > 1 shrinks image, geometry and target alike (the emulator's cases)."""
    from vello_amd import BlendMode, Compose, ImageData, Mix
    iw, ih = 640 // scale, 480 // scale
    yy, xx = np.mgrid[0:ih, 0:iw].astype(np.float64)
    r = np.hypot((xx - iw / 2.0) / (iw / 2.0), (yy - ih / 2.0) / (ih / 2.0))
    lum = np.clip(1.0 - r, 0.0, 1.0)
    stripe = ((xx + 2.0 * yy) // max(1, 40 // scale)).astype(np.int64) % 3
    px = np.zeros((ih, iw, 4), dtype=np.uint8)
    for c in range(3):
        px[..., c] = np.round(255.0 * lum * np.where(stripe == c, 1.0, 0.55))
    px[..., 3] = 255
    image = ImageData(px)
    k = 1.0 / scale
    s = Scene()
    s.push_layer(Fill.NonZero, BlendMode(Mix.Normal, Compose.SrcOver), 1.0, Affine.IDENTITY, Rect(0., 0., 700. * k, 500. * k))
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(245, 245, 220), None, Rect(0., 0., 640. * k, 240. * k))    # css BEIGE
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(127, 255, 212), None, Rect(0., 240. * k, 320. * k, 480. * k))  # css AQUAMARINE
    s.push_luminance_mask_layer(Fill.NonZero, 1.0, Affine.IDENTITY, Rect(0., 0., 640. * k, 480. * k))
    s.draw_image(image, Affine.IDENTITY)
    s.pop_layer()
    s.pop_layer()
    return s, 700 // scale, 500 // scale


def base_color_test_scene():
    """test_scenes.rs:1693-1706: a half-transparent white square over the BASE colour (which the scene animates through the hues: the
    tests render it over several).  550 x 550."""
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color(1.0, 1.0, 1.0, 0.5), None, Rect(50.0, 50.0, 500.0, 500.0))
    return s, 550, 550


def image_extend_modes_scene(quality=None):
    """test_scenes.rs:2168-2212: the 2x2 sample image as the brush of a 6x6 rect magnified 100x, brush offset (2, 2),
    under Pad, Reflect, Repeat and mixed x-Repeat / y-Reflect; bilinear or nearest.  1500 x 1500 on white."""
    from vello_amd import Extend, ImageBrush, ImageData, ImageQuality
    quality = ImageQuality.Medium if quality is None else quality
    px = np.zeros((2, 2, 4), dtype=np.uint8)
    px[0, 0], px[0, 1], px[1, 0], px[1, 1] = _rgba(255, 0, 0, 255), _rgba(0, 0, 255, 255), _rgba(0, 255, 255, 255), _rgba(255, 0, 255, 255)
    image = ImageData(px)
    off = Affine.translate(2., 2.)
    rect = Rect(0., 0., 6., 6.)
    s = Scene()
    for (xe, ye), (tx, ty) in [((Extend.Pad, Extend.Pad), (100., 100.)), ((Extend.Reflect, Extend.Reflect), (100., 800.)),
                               ((Extend.Repeat, Extend.Repeat), (800., 100.)), ((Extend.Repeat, Extend.Reflect), (800., 800.))]:
        s.fill(Fill.NonZero, Affine.scale(100.).then_translate(tx, ty), ImageBrush(image, xe, ye, quality), off, rect)
    return s, 1500, 1500


def brush_transform_scene(th=0.7):
    """test_scenes.rs:944-976 at one instant of its animation: a radial gradient under a rotated non-uniform scale, and a
    linear gradient whose BRUSH transform rotates about the shape's centre, filled and stroked.  1300 x 900."""
    import math
    from vello_amd import Gradient
    red, green, blue = Color.from_rgb8(255, 0, 0), Color.from_rgb8(0, 128, 0), Color.from_rgb8(0, 0, 255)
    linear = Gradient.new_linear((0.0, 0.0), (0.0, 200.0)).with_stops([red, green, blue])
    around = Affine.translate(200.0, 100.0) * Affine.rotate(th) * Affine.translate(-200.0, -100.0)
    s = Scene()
    s.fill(Fill.NonZero, Affine.rotate(math.radians(25.0)) * Affine.scale_non_uniform(2.0, 1.0),
           Gradient.new_radial((200.0, 200.0), 80.0).with_stops([red, green, blue]), None, Rect(100.0, 100.0, 300.0, 300.0))
    s.fill(Fill.NonZero, Affine.translate(200.0, 600.0), linear, around, Rect(0.0, 0.0, 400.0, 200.0))
    s.stroke(Stroke(40.0), Affine.translate(800.0, 600.0), linear, around, Rect(0.0, 0.0, 400.0, 200.0))
    return s, 1300, 900


def clip_blends_scene():
    """vello_tests/tests/known_issues.rs:54-90 (issue #1198): a Multiply layer inside a clip layer of the same triangle,
    over blue.  100 x 100."""
    from vello_amd import BlendMode, Compose, Mix
    s = Scene()
    box = Rect(0., 0., 100., 100.)
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(0, 0, 255), None, box)
    tri = BezPath()
    tri.move_to((50., 0.)); tri.line_to((0., 100.)); tri.line_to((100., 100.)); tri.close_path()
    s.push_clip_layer(Fill.NonZero, Affine.IDENTITY, tri)
    s.push_layer(Fill.NonZero, BlendMode(Mix.Multiply, Compose.SrcOver), 1.0, Affine.IDENTITY, tri)
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(127, 255, 212), None, box)
    s.pop_layer()
    s.pop_layer()
    return s, 100, 100


def regression_stroke_scenes():
    """vello_tests/tests/regression.rs:18-31 (issue #616: the 2 px stroke of a rounded rectangle must be watertight) and
    :107-121 (issue #662: a zero-width stroke draws nothing); known_issues.rs:20-52 (below).  Returns [(scene, w, h, name)]."""
    from vello_amd import RoundedRect
    a = Scene()
    a.stroke(Stroke(2.0), Affine.IDENTITY, Color.from_rgb8(255, 255, 255), None, RoundedRect(60.0, 10.0, 80.0, 30.0, 10.0))
    b = Scene()
    b.stroke(Stroke(0.0), Affine.IDENTITY, Color.from_rgb8(255, 218, 185), None, Rect(10.0, 10.0, 40.0, 40.0))
    # known_issues.rs:20-52 (issue #1061, `should_panic` upstream: the snapshot and the pipeline disagree about it): a Compose::Clear
    # layer WITHOUT content over a red square.  Whatever the reference's shaders make of it, the engine must make the same (the
    # oracle restates them: the layer's rectangle comes out cleared).
    from vello_amd import BlendMode, Compose, Mix
    c = Scene()
    c.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(0, 255, 0), None, Rect(0.0, 0.0, 60.0, 60.0))
    c.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(255, 0, 0), None, Rect(20.0, 20.0, 40.0, 40.0))
    c.push_layer(Fill.NonZero, BlendMode(Mix.Normal, Compose.Clear), 1.0, Affine.IDENTITY, Rect(20.0, 20.0, 40.0, 40.0))
    c.pop_layer()
    return [(a, 70, 30, "rounded_rectangle_watertight"), (b, 50, 50, "stroke_width_zero"), (c, 60, 60, "layer_size")]
