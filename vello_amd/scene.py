"""vello::Scene (vello/src/scene.rs:45-470) over the C++ host mirror."""
import ctypes
import enum

import numpy as np

from ._lib import load_library, LayoutStruct
from .kurbo import Affine


class Fill(enum.IntEnum):
    NonZero = 0
    EvenOdd = 1


class Mix(enum.IntEnum):
    Normal = 0; Multiply = 1; Screen = 2; Overlay = 3; Darken = 4; Lighten = 5; ColorDodge = 6; ColorBurn = 7
    HardLight = 8; SoftLight = 9; Difference = 10; Exclusion = 11; Hue = 12; Saturation = 13; Color = 14
    Luminosity = 15; Clip = 128


class Compose(enum.IntEnum):
    Clear = 0; Copy = 1; Dest = 2; SrcOver = 3; DestOver = 4; SrcIn = 5; DestIn = 6; SrcOut = 7; DestOut = 8
    SrcAtop = 9; DestAtop = 10; Xor = 11; Plus = 12; PlusLighter = 13


class BlendMode:
    def __init__(self, mix=Mix.Normal, compose=Compose.SrcOver):
        self.mix, self.compose = int(mix), int(compose)


class Color:
    """peniko::Color: sRGB, straight alpha, f32 components."""
    __slots__ = ("rgba",)

    def __init__(self, r, g, b, a=1.0):
        self.rgba = (np.float32(r), np.float32(g), np.float32(b), np.float32(a))

    @staticmethod
    def from_rgba8(r, g, b, a=255):
        f = np.float32(255.0)
        return Color(np.float32(r) / f, np.float32(g) / f, np.float32(b) / f, np.float32(a) / f)

    from_rgb8 = from_rgba8

    def with_alpha(self, a):
        return Color(self.rgba[0], self.rgba[1], self.rgba[2], a)

    def _ptr(self):
        return (ctypes.c_float * 4)(*[float(v) for v in self.rgba])

    def premul_rgba8(self):
        return load_library().vh_color_premul_rgba8(self._ptr())


def _path_args(shape):
    path = shape.path_elements(0.1)
    v, c = path.arrays()
    return v, c, v.ctypes.data, c.ctypes.data, len(v)


class Scene:
    def __init__(self):
        self._lib = load_library()
        self._h = self._lib.vh_scene_new()

    def __del__(self):
        try:
            if self._h:
                self._lib.vh_scene_free(self._h)
        except Exception:
            pass
        self._h = None

    def reset(self):
        self._lib.vh_scene_reset(self._h)

    def fill(self, style, transform, brush, brush_transform, shape):
        """Scene::fill (scene.rs:316-340); solid-colour brushes only, brush_transform must be None."""
        assert brush_transform is None, "brush transforms only matter for gradient/image brushes (out of scope)"
        v, c, vp, cp, n = _path_args(shape)
        self._lib.vh_scene_fill(self._h, int(style), transform._ptr(), brush._ptr(), vp, cp, n)

    def stroke(self, style, transform, brush, brush_transform, shape):
        """Scene::stroke (scene.rs:347-440), GPU stroker path."""
        assert brush_transform is None
        v, c, vp, cp, n = _path_args(shape)
        r = self._lib.vh_scene_stroke(self._h, style.width, int(style.join), style.miter_limit, int(style.start_cap),
                                      int(style.end_cap), transform._ptr(), brush._ptr(), vp, cp, n)
        if r != 0:
            raise NotImplementedError("dashed strokes are expanded by kurbo::dash upstream; not restated")

    def push_layer(self, clip_style, blend, alpha, transform, clip):
        v, c, vp, cp, n = _path_args(clip)
        if not isinstance(blend, BlendMode):
            blend = BlendMode(blend)
        self._lib.vh_scene_push_layer(self._h, int(clip_style), blend.mix, blend.compose, alpha, transform._ptr(), vp, cp, n)

    def push_luminance_mask_layer(self, clip_style, alpha, transform, clip):
        v, c, vp, cp, n = _path_args(clip)
        self._lib.vh_scene_push_luminance_mask_layer(self._h, int(clip_style), alpha, transform._ptr(), vp, cp, n)

    def push_clip_layer(self, clip_style, transform, clip):
        v, c, vp, cp, n = _path_args(clip)
        self._lib.vh_scene_push_clip_layer(self._h, int(clip_style), transform._ptr(), vp, cp, n)

    def pop_layer(self):
        self._lib.vh_scene_pop_layer(self._h)

    def append(self, other, transform=None):
        self._lib.vh_scene_append(self._h, other._h, transform._ptr() if transform is not None else None)

    # --- encoding access (vello_encoding::Encoding) ---
    _STREAMS = {"path_tags": (0, np.uint8), "path_data": (1, np.uint32), "draw_tags": (2, np.uint32),
                "draw_data": (3, np.uint32), "transforms": (4, np.float32), "styles": (5, np.uint32)}

    def stream(self, name):
        which, dt = self._STREAMS[name]
        n = self._lib.vh_scene_stream_bytes(self._h, which)
        out = np.zeros(n // np.dtype(dt).itemsize, dtype=dt)
        if n:
            self._lib.vh_scene_stream_copy(self._h, which, out.ctypes.data)
        return out

    def counts(self):
        out = (ctypes.c_uint32 * 4)()
        self._lib.vh_scene_counts(self._h, out)
        return {"n_paths": out[0], "n_path_segments": out[1], "n_clips": out[2], "n_open_clips": out[3]}

    def resolve(self):
        """Resolver::resolve for solid-colour scenes -> (packed bytes as np.uint8, Layout)."""
        from .renderer import Layout
        ptr = ctypes.c_void_p()
        lay = (ctypes.c_uint32 * 10)()
        n = self._lib.vh_scene_resolve(self._h, ctypes.byref(ptr), lay)
        if n == 0:
            packed = np.zeros(0, dtype=np.uint8)
        else:
            packed = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
        return packed, Layout(*list(lay))
