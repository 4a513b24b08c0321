"""vello::Scene (vello/src/scene.rs:45-470) over the C++ host mirror."""
import ctypes
import enum

import numpy as np

from ._lib import load_library, LayoutStruct
from .kurbo import Affine, Stroke


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


class Extend(enum.IntEnum):
    Pad = 0; Repeat = 1; Reflect = 2


class InterpolationAlphaSpace(enum.IntEnum):
    Premultiplied = 0; Unpremultiplied = 1


class ImageFormat(enum.IntEnum):
    Rgba8 = 0; Bgra8 = 1


class ImageAlphaType(enum.IntEnum):
    Alpha = 0; AlphaPremultiplied = 1


class ImageQuality(enum.IntEnum):
    Low = 0; Medium = 1; High = 2


class _BrushHandle:
    """Owns a vello::Brush on the C++ side."""

    def __init__(self, h):
        self._lib = load_library()
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._lib.vh_brush_free(self._h)
        except Exception:
            pass
        self._h = None


class Gradient:
    """peniko::Gradient: new_linear / new_radial / new_two_point_radial / new_sweep + with_stops / with_extend."""

    def __init__(self, kind, params):
        self.kind, self.params = kind, [float(v) for v in params]
        self.extend = Extend.Pad
        self.interpolation_alpha_space = InterpolationAlphaSpace.Premultiplied
        self.stops = []

    @staticmethod
    def new_linear(p0, p1):
        return Gradient(0, [p0[0], p0[1], p1[0], p1[1]])

    @staticmethod
    def new_radial(center, radius):
        return Gradient(1, [center[0], center[1], center[0], center[1], 0.0, radius])

    @staticmethod
    def new_two_point_radial(c0, r0, c1, r1):
        return Gradient(1, [c0[0], c0[1], c1[0], c1[1], r0, r1])

    @staticmethod
    def new_sweep(center, start_angle, end_angle):
        return Gradient(2, [center[0], center[1], start_angle, end_angle])

    def with_stops(self, stops):
        """stops: colours (evenly spaced offsets, as peniko's ColorStopsSource for slices) or (offset, Color) pairs."""
        out = []
        n = len(stops)
        for i, st in enumerate(stops):
            if isinstance(st, Color):
                out.append((np.float32(i / (n - 1) if n > 1 else 0.0), st))
            else:
                out.append((np.float32(st[0]), st[1]))
        self.stops = out
        return self

    def with_extend(self, extend):
        self.extend = Extend(extend)
        return self

    def with_interpolation_alpha_space(self, space):
        self.interpolation_alpha_space = InterpolationAlphaSpace(space)
        return self

    def _handle(self):
        lib = load_library()
        p = (ctypes.c_double * 6)(*(self.params + [0.0] * (6 - len(self.params))))
        flat = []
        for off, col in self.stops:
            flat += [float(off)] + [float(v) for v in col.rgba]
        arr = (ctypes.c_float * max(len(flat), 1))(*flat)
        return _BrushHandle(lib.vh_brush_gradient(self.kind, p, int(self.extend), int(self.interpolation_alpha_space), arr, len(self.stops)))


_next_image_id = [1]


class ImageData:
    """peniko::ImageData: `pixels` is an HxWx4 uint8 array in `format` channel order."""

    def __init__(self, pixels, format=ImageFormat.Rgba8, alpha_type=ImageAlphaType.Alpha):
        self.pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        assert self.pixels.ndim == 3 and self.pixels.shape[2] == 4
        self.height, self.width = self.pixels.shape[:2]
        self.format, self.alpha_type = ImageFormat(format), ImageAlphaType(alpha_type)
        self.id = _next_image_id[0]
        _next_image_id[0] += 1


class ImageBrush:
    """peniko::ImageBrush = ImageData + ImageSampler {x_extend, y_extend, quality, alpha}."""

    def __init__(self, image, x_extend=Extend.Pad, y_extend=Extend.Pad, quality=ImageQuality.Medium, alpha=1.0):
        self.image = image
        self.x_extend, self.y_extend, self.quality, self.alpha = Extend(x_extend), Extend(y_extend), ImageQuality(quality), float(alpha)

    def _handle(self):
        im = self.image
        return _BrushHandle(load_library().vh_brush_image(im.id, im.width, im.height, int(im.format), int(im.alpha_type), im.pixels.ctypes.data,
                                                          int(self.x_extend), int(self.y_extend), int(self.quality), self.alpha))


def _brush_handle(brush):
    if isinstance(brush, Color):
        return _BrushHandle(load_library().vh_brush_solid(brush._ptr()))
    if isinstance(brush, ImageData):
        brush = ImageBrush(brush)
    return brush._handle()


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
        """Scene::fill (scene.rs:316-340); brush: Color, Gradient, ImageBrush or ImageData."""
        v, c, vp, cp, n = _path_args(shape)
        if isinstance(brush, Color) and brush_transform is None:
            self._lib.vh_scene_fill(self._h, int(style), transform._ptr(), brush._ptr(), vp, cp, n)
            return
        bh = _brush_handle(brush)
        self._lib.vh_scene_fill_brush(self._h, int(style), transform._ptr(), bh._h,
                                      brush_transform._ptr() if brush_transform is not None else None, vp, cp, n)

    def stroke(self, style, transform, brush, brush_transform, shape):
        """Scene::stroke (scene.rs:347-440), GPU stroker path."""
        v, c, vp, cp, n = _path_args(shape)
        if isinstance(brush, Color) and brush_transform is None:
            r = self._lib.vh_scene_stroke(self._h, style.width, int(style.join), style.miter_limit, int(style.start_cap),
                                          int(style.end_cap), transform._ptr(), brush._ptr(), vp, cp, n)
        else:
            bh = _brush_handle(brush)
            r = self._lib.vh_scene_stroke_brush(self._h, style.width, int(style.join), style.miter_limit, int(style.start_cap),
                                                int(style.end_cap), transform._ptr(), bh._h,
                                                brush_transform._ptr() if brush_transform is not None else None, vp, cp, n)
        if r != 0:
            raise NotImplementedError("dashed strokes are expanded by kurbo::dash upstream; not restated")

    def draw_blurred_rounded_rect(self, transform, rect, brush, radius, std_dev):
        """Scene::draw_blurred_rounded_rect (scene.rs:256-270); rect = (x0, y0, x1, y1)."""
        r = (ctypes.c_double * 4)(*[float(v) for v in rect])
        self._lib.vh_scene_draw_blurred_rounded_rect(self._h, transform._ptr(), r, brush._ptr(), float(radius), float(std_dev))

    def draw_blurred_rounded_rect_in(self, shape, transform, rect, brush, radius, std_dev):
        """Scene::draw_blurred_rounded_rect_in (scene.rs:282-309)."""
        v, c, vp, cp, n = _path_args(shape)
        r = (ctypes.c_double * 4)(*[float(x) for x in rect])
        self._lib.vh_scene_draw_blurred_rounded_rect_in(self._h, vp, cp, n, transform._ptr(), r, brush._ptr(), float(radius), float(std_dev))

    def draw_image(self, image, transform):
        """Scene::draw_image (scene.rs:443-452)."""
        bh = _brush_handle(image)
        self._lib.vh_scene_draw_image(self._h, bh._h, transform._ptr())

    def _push_stroked(self, kind, st, mix, compose, alpha, transform, clip):
        v, c, vp, cp, n = _path_args(clip)
        if getattr(st, "dash_pattern", None):
            raise NotImplementedError("dashed strokes are expanded by kurbo::dash upstream; not restated")
        r = self._lib.vh_scene_push_layer_stroked(self._h, kind, st.width, int(st.join), st.miter_limit, int(st.start_cap),
                                                  int(st.end_cap), mix, compose, alpha, transform._ptr(), vp, cp, n)
        if r != 0:
            raise NotImplementedError("dashed strokes are expanded by kurbo::dash upstream; not restated")

    def push_layer(self, clip_style, blend, alpha, transform, clip):
        """Scene::push_layer (scene.rs:105-121); clip_style is a Fill rule or a Stroke (clip to the stroked outline)."""
        if not isinstance(blend, BlendMode):
            blend = BlendMode(blend)
        if isinstance(clip_style, Stroke):
            return self._push_stroked(0, clip_style, blend.mix, blend.compose, alpha, transform, clip)
        v, c, vp, cp, n = _path_args(clip)
        self._lib.vh_scene_push_layer(self._h, int(clip_style), blend.mix, blend.compose, alpha, transform._ptr(), vp, cp, n)

    def push_luminance_mask_layer(self, clip_style, alpha, transform, clip):
        if isinstance(clip_style, Stroke):
            return self._push_stroked(1, clip_style, 0, 0, alpha, transform, clip)
        v, c, vp, cp, n = _path_args(clip)
        self._lib.vh_scene_push_luminance_mask_layer(self._h, int(clip_style), alpha, transform._ptr(), vp, cp, n)

    def push_clip_layer(self, clip_style, transform, clip):
        if isinstance(clip_style, Stroke):
            return self._push_stroked(2, clip_style, 0, 0, 1.0, transform, clip)
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
        """Resolver::resolve for solid-colour scenes -> (packed bytes as np.uint8, Layout).  Scenes with gradient or
        image brushes carry late-bound resources: use a `Resolver`."""
        from .renderer import Layout
        if self._lib.vh_scene_n_patches(self._h) != 0:
            raise ValueError("scene has ramp/image patches: resolve it with vello_amd.Resolver().resolve(scene)")
        ptr = ctypes.c_void_p()
        lay = (ctypes.c_uint32 * 10)()
        n = self._lib.vh_scene_resolve(self._h, ctypes.byref(ptr), lay)
        if n == 0:
            packed = np.zeros(0, dtype=np.uint8)
        else:
            packed = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
        return packed, Layout(*list(lay))


class Resolved:
    """What Resolver::resolve returns (resolve.rs:172-180): packed scene + Layout, the ramp texture and the image atlas
    work list [(x, y, HxWx4 uint8 pixels)]."""

    def __init__(self, packed, layout, ramps, atlas_size, atlas_resized, uploads):
        self.packed, self.layout, self.ramps = packed, layout, ramps
        self.atlas_size, self.atlas_resized, self.uploads = atlas_size, atlas_resized, uploads

    def __iter__(self):  # (packed, layout) unpacking, like Scene.resolve()
        return iter((self.packed, self.layout))

    def atlas_image(self):
        """The full atlas as an array (for the oracle, which takes the atlas whole)."""
        if not self.atlas_size:
            return None
        a = np.zeros((self.atlas_size, self.atlas_size, 4), dtype=np.uint8)
        for x, y, px in self.uploads:
            a[y:y + px.shape[0], x:x + px.shape[1]] = px
        return a


class Resolver:
    """vello_encoding::Resolver: owns the ramp cache and the image atlas allocation across frames."""

    def __init__(self, atlas_sizes=None):
        """atlas_sizes: optional (initial side, maximum side) of the image atlas (default 1024, 8192)."""
        self._lib = load_library()
        self._h = (self._lib.vh_resolver_new() if atlas_sizes is None
                   else self._lib.vh_resolver_new_with_atlas_sizes(int(atlas_sizes[0]), int(atlas_sizes[1])))
        self._resident = {}

    def __del__(self):
        try:
            if self._h:
                self._lib.vh_resolver_free(self._h)
        except Exception:
            pass
        self._h = None

    def resolve(self, scene):
        from .renderer import Layout
        ptr, ramps_p = ctypes.c_void_p(), ctypes.c_void_p()
        lay = (ctypes.c_uint32 * 10)()
        info = (ctypes.c_uint32 * 5)()
        n = self._lib.vh_resolver_resolve(self._h, scene._h, ctypes.byref(ptr), lay, ctypes.byref(ramps_p), info)
        packed = (np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
                  if n else np.zeros(0, dtype=np.uint8))
        n_ramps, atlas_size, atlas_resized, n_uploads = info[0], info[1], bool(info[2]), info[3]
        ramps = None
        if n_ramps:
            ramps = np.ctypeslib.as_array(ctypes.cast(ramps_p, ctypes.POINTER(ctypes.c_uint32)), shape=(n_ramps * 512,)).copy()
        if atlas_resized:
            self._resident = {}
        for i in range(n_uploads):
            xywh = (ctypes.c_uint32 * 4)()
            dp = self._lib.vh_resolver_upload(self._h, i, xywh)
            x, y, w, h = list(xywh)
            px = np.ctypeslib.as_array(ctypes.cast(dp, ctypes.POINTER(ctypes.c_uint8)), shape=(h, w, 4)).copy()
            self._resident.pop((x, y), None)  # a later upload paints over earlier ones: keep the list in upload order
            self._resident[(x, y)] = px
        uploads_all = [(x, y, px) for (x, y), px in self._resident.items()]
        r = Resolved(packed, Layout(*list(lay)), ramps, atlas_size, atlas_resized, uploads_all)
        r.new_uploads = n_uploads
        r.evicted = int(info[4])
        return r

    def mark_image_dirty(self, image):
        """Resolver::mark_image_dirty (resolve.rs:173-179): `image` (an ImageData) changed in place; the next resolve
        that uses it uploads it again."""
        self._lib.vh_resolver_mark_image_dirty(self._h, ctypes.c_uint64(image.id))

    def image_cache_info(self):
        """(resident images, atlas side): test / debug view of the image cache."""
        out = (ctypes.c_uint32 * 2)()
        self._lib.vh_resolver_image_cache_info(self._h, out)
        return int(out[0]), int(out[1])
