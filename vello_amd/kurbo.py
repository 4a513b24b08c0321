"""Python face of the kurbo subset in csrc/host/kurbo.{hpp,cpp} (Affine, BezPath, shapes, Stroke)."""
import ctypes
import enum
import math

import numpy as np

from ._lib import load_library

MOVE_TO, LINE_TO, QUAD_TO, CURVE_TO, CLOSE_PATH = 0, 1, 2, 3, 4
_NCOORD = {MOVE_TO: 2, LINE_TO: 2, QUAD_TO: 4, CURVE_TO: 6, CLOSE_PATH: 0}


class Affine:
    """kurbo::Affine: coefficients [a b c d e f], (x, y) -> (a x + c y + e, b x + d y + f)."""
    __slots__ = ("c",)

    def __init__(self, coeffs=(1.0, 0.0, 0.0, 1.0, 0.0, 0.0)):
        self.c = tuple(float(v) for v in coeffs)

    IDENTITY = None

    @staticmethod
    def translate(x, y):
        return Affine((1, 0, 0, 1, x, y))

    @staticmethod
    def scale(s):
        return Affine((s, 0, 0, s, 0, 0))

    @staticmethod
    def scale_non_uniform(sx, sy):
        return Affine((sx, 0, 0, sy, 0, 0))

    @staticmethod
    def rotate(th):
        s, c = math.sin(th), math.cos(th)
        return Affine((c, s, -s, c, 0, 0))

    @staticmethod
    def skew(skew_x, skew_y):
        """kurbo::Affine::skew: x' = x + skew_x * y, y' = skew_y * x + y."""
        return Affine((1, skew_y, skew_x, 1, 0, 0))

    # kurbo's then_* builders: the new map is applied AFTER self
    def then_translate(self, x, y):
        return Affine.translate(x, y) * self

    def then_rotate(self, th):
        return Affine.rotate(th) * self

    def then_scale(self, s):
        return Affine.scale(s) * self

    def then_scale_non_uniform(self, sx, sy):
        return Affine.scale_non_uniform(sx, sy) * self

    def __mul__(self, o):
        a, b = self.c, o.c
        return Affine((a[0] * b[0] + a[2] * b[1], a[1] * b[0] + a[3] * b[1], a[0] * b[2] + a[2] * b[3],
                       a[1] * b[2] + a[3] * b[3], a[0] * b[4] + a[2] * b[5] + a[4], a[1] * b[4] + a[3] * b[5] + a[5]))

    def _ptr(self):
        return (ctypes.c_double * 6)(*self.c)


Affine.IDENTITY = Affine()


class BezPath:
    """kurbo::BezPath as two flat arrays: verbs (u8) and coordinates (f64)."""

    def __init__(self, verbs=None, coords=None):
        self._verbs = list(verbs) if verbs is not None and not isinstance(verbs, np.ndarray) else verbs
        self._coords = list(coords) if coords is not None and not isinstance(coords, np.ndarray) else coords
        if self._verbs is None:
            self._verbs, self._coords = [], []

    def _lists(self):
        if isinstance(self._verbs, np.ndarray):
            self._verbs = self._verbs.tolist()
            self._coords = self._coords.tolist()

    def move_to(self, p):
        self._lists(); self._verbs.append(MOVE_TO); self._coords.extend(p)

    def line_to(self, p):
        self._lists(); self._verbs.append(LINE_TO); self._coords.extend(p)

    def quad_to(self, p1, p2):
        self._lists(); self._verbs.append(QUAD_TO); self._coords.extend(p1); self._coords.extend(p2)

    def curve_to(self, p1, p2, p3):
        self._lists(); self._verbs.append(CURVE_TO); self._coords.extend(p1); self._coords.extend(p2); self._coords.extend(p3)

    def close_path(self):
        self._lists(); self._verbs.append(CLOSE_PATH)

    def arrays(self):
        v = np.ascontiguousarray(self._verbs, dtype=np.uint8)
        c = np.ascontiguousarray(self._coords, dtype=np.float64)
        return v, c

    @staticmethod
    def from_arrays(verbs, coords):
        return BezPath(np.ascontiguousarray(verbs, dtype=np.uint8), np.ascontiguousarray(coords, dtype=np.float64))

    @staticmethod
    def _from_handle(h):
        lib = load_library()
        if not h:
            raise ValueError("invalid path")
        n, k = lib.vh_bezpath_n_verbs(h), lib.vh_bezpath_n_coords(h)
        v = np.zeros(max(n, 1), dtype=np.uint8)
        c = np.zeros(max(k, 1), dtype=np.float64)
        lib.vh_bezpath_copy(h, v.ctypes.data, c.ctypes.data)
        lib.vh_bezpath_free(h)
        return BezPath(v[:n], c[:k])

    @staticmethod
    def from_svg(d):
        """BezPath::from_svg (examples/scenes/src/pico_svg.rs:167)."""
        h = load_library().vh_bezpath_from_svg(d.encode())
        if not h:
            raise ValueError("SVG path syntax error")
        return BezPath._from_handle(h)

    def path_elements(self, tolerance=0.1):
        return self


class Circle:
    def __init__(self, center, radius):
        self.center, self.radius = center, radius

    def path_elements(self, tolerance=0.1):
        return BezPath._from_handle(load_library().vh_bezpath_circle(self.center[0], self.center[1], self.radius, tolerance))


class Rect:
    def __init__(self, x0, y0, x1, y1):
        self.x0, self.y0, self.x1, self.y1 = x0, y0, x1, y1

    @staticmethod
    def from_center_size(center, size):
        return Rect(center[0] - 0.5 * size[0], center[1] - 0.5 * size[1], center[0] + 0.5 * size[0], center[1] + 0.5 * size[1])

    def path_elements(self, tolerance=0.1):
        return BezPath._from_handle(load_library().vh_bezpath_rect(self.x0, self.y0, self.x1, self.y1))


class RoundedRect:
    def __init__(self, x0, y0, x1, y1, radius):
        self.r = (x0, y0, x1, y1, radius)

    def path_elements(self, tolerance=0.1):
        return BezPath._from_handle(load_library().vh_bezpath_rounded_rect(*self.r, tolerance))


class Line:
    def __init__(self, p0, p1):
        self.p0, self.p1 = p0, p1

    def path_elements(self, tolerance=0.1):
        return BezPath._from_handle(load_library().vh_bezpath_line(self.p0[0], self.p0[1], self.p1[0], self.p1[1]))


class Join(enum.IntEnum):
    Bevel = 0
    Miter = 1
    Round = 2


class Cap(enum.IntEnum):
    Butt = 0
    Square = 1
    Round = 2


class Stroke:
    """kurbo::Stroke; Stroke(width) has kurbo's defaults: round join, round caps, miter limit 4."""

    def __init__(self, width, join=Join.Round, miter_limit=4.0, start_cap=Cap.Round, end_cap=Cap.Round):
        self.width, self.join, self.miter_limit, self.start_cap, self.end_cap = width, join, miter_limit, start_cap, end_cap

    def with_caps(self, cap):
        return Stroke(self.width, self.join, self.miter_limit, cap, cap)

    def with_join(self, join):
        return Stroke(self.width, join, self.miter_limit, self.start_cap, self.end_cap)

    def with_miter_limit(self, limit):
        return Stroke(self.width, self.join, limit, self.start_cap, self.end_cap)
