"""Seeded random scenes over the WHOLE scene API for differential testing (kernels vs oracle): every brush kind with
degenerate parameters, layers with every mix / compose mode, luminance masks, fill- and stroke-styled clips, strokes
with zero / hairline / huge widths, affines with skew, reflection and near-singular scale, repeated points, far
off-screen and sub-pixel geometry.  Harness code, not a product feature."""
import math

import numpy as np

from vello_amd import (Affine, BezPath, BlendMode, Cap, Circle, Color, Compose, Extend, Fill, Gradient, ImageAlphaType, ImageBrush,
                       ImageData, ImageFormat, ImageQuality, InterpolationAlphaSpace, Join, Mix, Rect, Scene, Stroke)

_MIXES = [m for m in Mix if m != Mix.Clip]
_COMPOSES = list(Compose)


def _color(rng):
    a = float(rng.choice([1.0, 1.0, 0.0, 0.5, rng.uniform(0, 1)]))
    return Color(float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), a)


def _affine(rng, size):
    k = int(rng.integers(0, 8))
    if k < 3:
        return Affine.IDENTITY
    c = Affine.translate(size / 2, size / 2)
    ci = Affine.translate(-size / 2, -size / 2)
    if k == 3:
        return c * Affine.rotate(float(rng.uniform(0, 2 * math.pi))) * ci
    if k == 4:
        return c * Affine.scale_non_uniform(float(rng.uniform(-1.5, 1.5)), float(rng.uniform(0.2, 1.5))) * ci
    if k == 5:
        return c * Affine.skew(float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-0.5, 0.5))) * ci
    if k == 6:
        return c * Affine.scale_non_uniform(float(rng.choice([1e-3, 1e-6, 1.0])), 1.0) * Affine.rotate(float(rng.uniform(0, 3))) * ci
    return Affine([float(v) for v in rng.uniform(-1.5, 1.5, 4)] + [float(rng.uniform(0, size)), float(rng.uniform(0, size))])


_EXTREME = False  # fuzz_scene(..., extreme=True): also coordinates far beyond the viewport, up to f32's limits, inf and NaN
_EXTREME_VALUES = [1e6, -1e6, 3e7, -2e8, 1e9, 1e12, 1e20, 3e38, -3e38, float("inf"), float("-inf"), float("nan")]


def _point(rng, size):
    if _EXTREME and rng.random() < 0.12:
        pick = lambda: float(rng.choice(_EXTREME_VALUES)) if rng.random() < 0.7 else float(rng.uniform(-10, size + 10))
        return (pick(), pick())
    k = int(rng.integers(0, 10))
    if k == 0:
        return (float(rng.uniform(-1e4, 1e4)), float(rng.uniform(-1e4, 1e4)))     # far outside the viewport
    if k == 1:
        return (float(rng.integers(0, size // 16 + 1) * 16), float(rng.integers(0, size // 16 + 1) * 16))  # on tile corners
    return (float(rng.uniform(-10, size + 10)), float(rng.uniform(-10, size + 10)))


def _shape(rng, size):
    k = int(rng.integers(0, 8))
    if k == 0:
        p = _point(rng, size)
        return Circle(p, float(rng.choice([0.0, 0.3, rng.uniform(1, size / 2)])))
    if k == 1:
        a, b = _point(rng, size), _point(rng, size)
        return Rect(a[0], a[1], b[0], b[1])
    p = BezPath()
    last = _point(rng, size)
    p.move_to(last)
    for _ in range(int(rng.integers(1, 9))):
        v = int(rng.integers(0, 7))
        if v == 0:
            p.line_to(last)                                   # zero-length segment
        elif v == 1:
            last = _point(rng, size); p.line_to(last)
        elif v == 2:
            c = _point(rng, size); last = _point(rng, size); p.quad_to(c, last)
        elif v == 3:
            c0, c1 = _point(rng, size), _point(rng, size); last = _point(rng, size); p.curve_to(c0, c1, last)
        elif v == 4:
            p.curve_to(last, last, last)                      # fully degenerate cubic
        elif v == 5:
            q = (last[0] + float(rng.uniform(-0.01, 0.01)), last[1] + float(rng.uniform(-0.01, 0.01)))
            p.line_to(q); last = q                            # sub-pixel step
        else:
            p.close_path(); last = _point(rng, size); p.move_to(last)
    if rng.random() < 0.5:
        p.close_path()
    return p


def _gradient(rng, size):
    n = int(rng.choice([0, 1, 2, 2, 3, 5]))
    stops = sorted(float(v) for v in rng.uniform(0, 1, n))
    if n >= 2 and rng.random() < 0.3:
        stops[1] = stops[0]                                   # coincident stops
    cs = [(stops[i], _color(rng)) for i in range(n)]
    k = int(rng.integers(0, 4))
    if k == 0:
        a = _point(rng, size)
        b = a if rng.random() < 0.15 else _point(rng, size)   # zero-length axis
        g = Gradient.new_linear(a, b)
    elif k == 1:
        g = Gradient.new_radial(_point(rng, size), float(rng.choice([0.0, rng.uniform(1, size / 2)])))
    elif k == 2:
        c0 = _point(rng, size)
        c1 = c0 if rng.random() < 0.3 else _point(rng, size)
        r0 = float(rng.choice([0.0, rng.uniform(0, size / 3)]))
        r1 = r0 if rng.random() < 0.3 else float(rng.uniform(0, size / 2))
        g = Gradient.new_two_point_radial(c0, r0, c1, r1)
    else:
        a0 = float(rng.uniform(-7, 7))
        g = Gradient.new_sweep(_point(rng, size), a0, a0 if rng.random() < 0.15 else float(rng.uniform(-7, 7)))
    g = g.with_stops(cs).with_extend(Extend(int(rng.integers(0, 3))))
    return g.with_interpolation_alpha_space(InterpolationAlphaSpace(int(rng.integers(0, 2))))


def _image(rng):
    w, h = int(rng.choice([1, 2, 3, 7, 16, 24])), int(rng.choice([1, 2, 5, 16, 19]))
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    if rng.random() < 0.4:
        px[:, :, 3] = 255
    at = ImageAlphaType(int(rng.integers(0, 2)))
    if at == ImageAlphaType.AlphaPremultiplied:
        px[:, :, :3] = (px[:, :, :3].astype(np.uint16) * px[:, :, 3:4] // 255).astype(np.uint8)
    im = ImageData(px, ImageFormat(int(rng.integers(0, 2))), at)
    return ImageBrush(im, Extend(int(rng.integers(0, 3))), Extend(int(rng.integers(0, 3))), ImageQuality(int(rng.integers(0, 3))),
                      float(rng.choice([1.0, 0.0, rng.uniform(0, 1)])))


def _brush(rng, size):
    k = int(rng.integers(0, 10))
    if k < 5:
        return _color(rng)
    if k < 8:
        return _gradient(rng, size)
    return _image(rng)


def _stroke(rng):
    w = float(rng.choice([0.0, 1e-3, 0.5, rng.uniform(0.5, 20.0), 80.0]))
    return Stroke(w, join=Join(int(rng.integers(0, 3))), miter_limit=float(rng.choice([0.0, 1.0, 4.0, 100.0])),
                  start_cap=Cap(int(rng.integers(0, 3))), end_cap=Cap(int(rng.integers(0, 3))))


def fuzz_scene(seed, size=128, n_ops=40, extreme=False):
    global _EXTREME
    _EXTREME = bool(extreme)
    try:
        return _fuzz_scene(seed, size, n_ops)
    finally:
        _EXTREME = False


def _fuzz_scene(seed, size, n_ops):
    rng = np.random.Generator(np.random.PCG64(seed))
    s = Scene()
    depth = 0
    for _ in range(int(rng.integers(1, n_ops))):
        op = int(rng.integers(0, 20))
        t = _affine(rng, size)
        if op < 8:
            bt = None if rng.random() < 0.6 else _affine(rng, size)
            s.fill(Fill(int(rng.integers(0, 2))), t, _brush(rng, size), bt, _shape(rng, size))
        elif op < 12:
            bt = None if rng.random() < 0.6 else _affine(rng, size)
            s.stroke(_stroke(rng), t, _brush(rng, size), bt, _shape(rng, size))
        elif op == 12:
            b = _image(rng)
            s.draw_image(b, t * Affine.scale(float(rng.uniform(0.5, 6.0))))
        elif op == 13:
            a, b = _point(rng, size), _point(rng, size)
            s.draw_blurred_rounded_rect(t, (min(a[0], b[0]), min(a[1], b[1]), max(a[0], b[0]), max(a[1], b[1])), _color(rng),
                                        float(rng.choice([0.0, rng.uniform(0, 30)])), float(rng.choice([0.0, 0.1, rng.uniform(0.5, 12)])))
        elif op < 18 and depth < 7:
            style = _stroke(rng) if rng.random() < 0.25 else Fill(int(rng.integers(0, 2)))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                s.push_clip_layer(style, t, _shape(rng, size))
            elif kind == 1:
                s.push_luminance_mask_layer(style, float(rng.uniform(0, 1)), t, _shape(rng, size))
            else:
                bm = BlendMode(_MIXES[int(rng.integers(0, len(_MIXES)))], _COMPOSES[int(rng.integers(0, len(_COMPOSES)))])
                s.push_layer(style, bm, float(rng.choice([1.0, 0.0, rng.uniform(0, 1)])), t, _shape(rng, size))
            depth += 1
        elif depth > 0:
            s.pop_layer()
            depth -= 1
    for _ in range(depth):
        s.pop_layer()
    return s
