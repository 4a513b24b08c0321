"""vello::Renderer / RenderParams / AaConfig (vello/src/lib.rs:175-193, :357-369, :432-515) and a
direct binding of the engine C ABI (include/vello_hip.h) for measurement and differential tests."""
import collections
import ctypes
import enum

import numpy as np

from ._lib import load_library, VelloHipError, Capacities, Bump, LayoutStruct, RenderParamsStruct
from .scene import Color


class AaConfig(enum.IntEnum):
    Area = 0
    Msaa8 = 1
    Msaa16 = 2


Layout = collections.namedtuple("Layout", [
    "n_draw_objects", "n_paths", "n_clips", "bin_data_start", "path_tag_base", "path_data_base", "draw_tag_base",
    "draw_data_base", "transform_base", "style_base"])

STAGES = ["pathtag_scan", "flatten", "draw_scan", "clip", "binning", "tile_alloc", "path_count", "backdrop", "coarse",
          "path_tiling", "fine"]
BUFFERS = ["scene", "config", "tag_monoids", "path_bboxes", "bump", "lines", "draw_monoids", "info_bin_data", "clip_inp",
           "clip_bboxes", "draw_bboxes", "bin_headers", "paths", "tiles", "seg_counts", "segments", "ptcl", "blend_spill",
           "output"]

E_CAPACITY = -4


class RenderParams:
    def __init__(self, base_color, width, height, antialiasing_method=AaConfig.Area):
        self.base_color, self.width, self.height, self.antialiasing_method = base_color, width, height, antialiasing_method


class RendererOptions:
    def __init__(self, device=0, antialiasing_support=7, capacities=None):
        self.device, self.antialiasing_support, self.capacities = device, antialiasing_support, capacities


def _caps(capacities):
    if capacities is None:
        return None
    c = Capacities()
    for k, v in capacities.items():
        setattr(c, k, v)
    return ctypes.byref(c)


def _data_ptr(texture, width=None, height=None):
    """Accepts a numpy array (host) or a torch tensor (host or device): dense uint8 rows of `width` RGBA8 pixels.
    With width / height given, the buffer must hold the whole target (the engine writes height * width * 4 bytes)."""
    if isinstance(texture, np.ndarray):
        if texture.dtype != np.uint8 or not texture.flags["C_CONTIGUOUS"]:
            raise ValueError("target must be a C-contiguous uint8 array")
        nbytes, ptr, is_dev = texture.nbytes, texture.ctypes.data, False
    else:
        import torch

        if texture.dtype != torch.uint8 or not texture.is_contiguous():
            raise ValueError("target must be a contiguous torch.uint8 tensor")
        nbytes, ptr, is_dev = texture.numel(), texture.data_ptr(), bool(texture.is_cuda)
    if width is not None and nbytes < int(width) * int(height) * 4:
        raise ValueError(f"target holds {nbytes} bytes, a {width}x{height} RGBA8 frame needs {int(width) * int(height) * 4}")
    return ptr, is_dev


class Renderer:
    """Renderer::new + Renderer::render_to_texture.  Raises VelloHipError when no GPU is usable."""

    def __init__(self, options=None):
        options = options or RendererOptions()
        self._lib = load_library()
        err = ctypes.create_string_buffer(512)
        self._h = self._lib.vh_renderer_new(options.device, options.antialiasing_support, _caps(options.capacities), err, 512)
        if not self._h:
            raise VelloHipError(err.value.decode() or "vello_hip_create failed")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.vh_renderer_free(self._h)
        except Exception:
            pass
        self._h = None

    def render_to_texture(self, scene, texture, params):
        ptr, is_dev = _data_ptr(texture, params.width, params.height)
        stride = params.width * 4
        r = self._lib.vh_renderer_render_to_texture(self._h, scene._h, ptr, stride, 1 if is_dev else 0, params.width,
                                                    params.height, params.base_color._ptr(), int(params.antialiasing_method))
        if r != 0:
            raise VelloHipError(f"render_to_texture failed ({r}): {self._lib.vh_renderer_error(self._h).decode()}")

    def last_bump(self):
        b = Bump()
        self._lib.vh_renderer_last_bump(self._h, ctypes.byref(b))
        return b.as_dict()


class Engine:
    """Direct binding of include/vello_hip.h (one context = one GPU, one stream)."""

    def __init__(self, device=0, aa_mask=7, capacities=None):
        self._lib = load_library()
        h = ctypes.c_void_p()
        r = self._lib.vello_hip_create(device, aa_mask, _caps(capacities), ctypes.byref(h))
        if r != 0:
            raise VelloHipError(f"vello_hip_create failed ({r}): {self._lib.vello_hip_last_error(None).decode()}")
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.vello_hip_destroy(self._h)
        except Exception:
            pass
        self._h = None

    def _check(self, r, what):
        if r != 0:
            raise VelloHipError(f"{what} failed ({r}): {self._lib.vello_hip_last_error(self._h).decode()}")

    @staticmethod
    def _params(width, height, base_color, aa):
        bc = base_color.premul_rgba8() if isinstance(base_color, Color) else int(base_color)
        return RenderParamsStruct(width, height, bc, int(aa))

    def upload_scene(self, packed, layout, ramps=None):
        """Command::Upload of the packed scene (+ the gradient ramp texture: n_ramps x 512 RGBA8 texels as uint32)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        lay = LayoutStruct(*layout)
        rp, nr = None, 0
        if ramps is not None and len(ramps):
            ramps = np.ascontiguousarray(ramps, dtype=np.uint32)
            rp, nr = ramps.ctypes.data, ramps.size // 512
        self._check(self._lib.vello_hip_upload_scene(self._h, packed.ctypes.data, packed.nbytes, ctypes.byref(lay), rp, nr), "upload_scene")

    def resize_image_atlas(self, width, height):
        self._check(self._lib.vello_hip_resize_image_atlas(self._h, width, height), "resize_image_atlas")

    def write_image(self, x, y, pixels):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        h, w = pixels.shape[:2]
        self._check(self._lib.vello_hip_write_image(self._h, x, y, w, h, pixels.ctypes.data, w * 4), "write_image")

    def upload_resolved(self, resolved):
        """Everything a Resolver result carries: atlas (re)creation + image writes, ramps, packed scene."""
        if resolved.atlas_size:
            self.resize_image_atlas(resolved.atlas_size, resolved.atlas_size)
            for x, y, px in resolved.uploads:
                self.write_image(x, y, px)
        self.upload_scene(resolved.packed, resolved.layout, resolved.ramps)

    def render_frame(self, packed, layout, width, height, base_color, aa, out=None, ramps=None):
        """vello_hip_render_frame: upload this frame's scene into the next in-flight slot and enqueue the frame."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        lay = LayoutStruct(*layout)
        p = self._params(width, height, base_color, aa)
        ptr, stride = None, 0
        if out is not None:
            ptr, is_dev = _data_ptr(out, width, height)
            assert is_dev, "render_frame writes to device memory"
            stride = width * 4
        rp, nr = None, 0
        if ramps is not None and len(ramps):
            ramps = np.ascontiguousarray(ramps, dtype=np.uint32)
            rp, nr = ramps.ctypes.data, ramps.size // 512
        self._check(self._lib.vello_hip_render_frame(self._h, packed.ctypes.data, packed.nbytes, ctypes.byref(lay), ctypes.byref(p),
                                                     rp, nr, ptr, stride), "render_frame")

    def render_resident(self, width, height, base_color, aa, out=None):
        p = self._params(width, height, base_color, aa)
        ptr, stride = None, 0
        if out is not None:
            ptr, is_dev = _data_ptr(out, width, height)
            assert is_dev, "render_resident writes to device memory"
            stride = width * 4
        self._check(self._lib.vello_hip_render_resident(self._h, ctypes.byref(p), ptr, stride), "render_resident")

    def render(self, packed, layout, width, height, base_color, aa, ramps=None):
        """One blocking frame; returns (HxWx4 uint8 image, bump dict)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        lay = LayoutStruct(*layout)
        p = self._params(width, height, base_color, aa)
        out = np.zeros((height, width, 4), dtype=np.uint8)
        b = Bump()
        rp, nr = None, 0
        if ramps is not None and len(ramps):
            ramps = np.ascontiguousarray(ramps, dtype=np.uint32)
            rp, nr = ramps.ctypes.data, ramps.size // 512
        r = self._lib.vello_hip_render(self._h, packed.ctypes.data, packed.nbytes, ctypes.byref(lay), ctypes.byref(p), rp, nr,
                                       out.ctypes.data, width * 4, 0, ctypes.byref(b))
        if r != 0 and r != E_CAPACITY:
            self._check(r, "render")
        return out, b.as_dict()

    def capacities(self):
        c = Capacities()
        self._check(self._lib.vello_hip_get_capacities(self._h, ctypes.byref(c)), "get_capacities")
        return {k: getattr(c, k) for k, _ in Capacities._fields_}

    def grow_pools(self, bump):
        """vello_hip_grow_pools with a bump dict; returns True if any pool grew."""
        b = Bump(*[bump[k] for k, _ in Bump._fields_])
        return self._lib.vello_hip_grow_pools(self._h, ctypes.byref(b), None) == 0

    def set_auto_grow(self, enabled=True):
        self._check(self._lib.vello_hip_set_auto_grow(self._h, 1 if enabled else 0), "set_auto_grow")

    def set_debug_flags(self, no_cull=False, stroke_kernel=False, seq_clip=False, fine_slices=False, flatten_coop=False, flatten_alone=False, no_fusion=False):
        """vello_hip_set_debug_flags: no_cull makes coarse emit every draw (reference-exact PTCL / segments); stroke_kernel
        runs flatten's stroked-line kernel whatever the number of stroked lines; seq_clip matches clips with the one-wave
        stack machine instead of the partitioned kernels; fine_slices cuts every tile's command list into slices of
        4 fills for fine's MSAA modes (normally only lists of >= 96 fills are cut, engine.h FINE_SLICE_MIN_FILLS); flatten_coop /
        flatten_alone pick the kernels of flatten's heavy list (the wave-cooperative walk / every lane on its own) instead of leaving
        the choice to the engine; no_fusion launches every stage of a small scene as a kernel of its own (normally consecutive stages
        up to tile_alloc share launches there).  Flags not named are cleared (update_debug_flags keeps them)."""
        self._debug = {"no_cull": bool(no_cull), "stroke_kernel": bool(stroke_kernel), "seq_clip": bool(seq_clip),
                       "fine_slices": bool(fine_slices), "flatten_coop": bool(flatten_coop), "flatten_alone": bool(flatten_alone),
                       "no_fusion": bool(no_fusion)}
        d = self._debug
        flags = ((1 if d["no_cull"] else 0) | (2 if d["stroke_kernel"] else 0) | (4 if d["seq_clip"] else 0) | (8 if d["fine_slices"] else 0) |
                 (16 if d["flatten_coop"] else 0) | (32 if d["flatten_alone"] else 0) | (64 if d["no_fusion"] else 0))
        self._check(self._lib.vello_hip_set_debug_flags(self._h, flags), "set_debug_flags")

    def update_debug_flags(self, **changes):
        """set_debug_flags with the flags not named left as they are."""
        d = dict(getattr(self, "_debug", {}))
        d.update(changes)
        self.set_debug_flags(**d)

    def last_render_attempts(self):
        return int(self._lib.vello_hip_last_render_attempts(self._h))

    def fused_launches(self):
        """vello_hip_fused_launches: launches in which stages of a small scene shared a kernel, since the engine was created."""
        return int(self._lib.vello_hip_fused_launches(self._h))

    def set_frames_in_flight(self, n):
        self._check(self._lib.vello_hip_set_frames_in_flight(self._h, n), "set_frames_in_flight")
        self.KERNELS = dict(self.KERNELS, flatten=self.FLATTEN_KERNELS[0 if n == 1 else 1])

    def stream(self):
        """vello_hip_get_stream: the hipStream_t (as an int) of the lane that rendered the newest frame."""
        return int(self._lib.vello_hip_get_stream(self._h) or 0)

    def sync_frame(self, age=0):
        self._check(self._lib.vello_hip_sync_frame(self._h, age), "sync_frame")

    def sync(self):
        return self._lib.vello_hip_sync(self._h)

    def bump(self):
        b = Bump()
        self._check(self._lib.vello_hip_get_bump(self._h, ctypes.byref(b)), "get_bump")
        return b.as_dict()

    def run_stages(self, width, height, base_color, aa, first, last):
        p = self._params(width, height, base_color, aa)
        first = STAGES.index(first) if isinstance(first, str) else first
        last = STAGES.index(last) if isinstance(last, str) else last
        self._check(self._lib.vello_hip_run_stages(self._h, ctypes.byref(p), first, last), "run_stages")

    def read_buffer(self, name, dtype=np.uint8, count_bytes=None, offset=0):
        bid = BUFFERS.index(name)
        size = self._lib.vello_hip_buffer_size(self._h, bid)
        n = size - offset if count_bytes is None else min(count_bytes, size - offset)
        out = np.zeros(n, dtype=np.uint8)
        self._check(self._lib.vello_hip_read_buffer(self._h, bid, out.ctypes.data, offset, n), f"read_buffer({name})")
        return out.view(dtype) if n % np.dtype(dtype).itemsize == 0 else out

    def control_words(self):
        """The first 64 words of the last frame's 128-word control block (engine.h Control; the rest are flatten's arc
        sub-list counters): bump allocators, tickets, flatten's list lengths, fine's bucket counters [16:48], slice items [48]
        and coverage-scratch words [49] handed out by coarse."""
        return self.read_buffer("bump", np.uint32, 256)

    def fine_slice_stats(self):
        """(slice items, coverage-scratch words) coarse handed out for the last frame."""
        w = self.control_words()
        return int(w[48]), int(w[49])

    def write_buffer(self, name, data, offset=0):
        bid = BUFFERS.index(name)
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self._check(self._lib.vello_hip_write_buffer(self._h, bid, data.ctypes.data, offset, data.nbytes), f"write_buffer({name})")

    def set_profiling(self, stages):
        mask = 0
        for s in stages:
            mask |= 1 << (STAGES.index(s) if isinstance(s, str) else s)
        self._lib.vello_hip_set_profiling(self._h, mask)

    def stage_ms(self):
        ms = (ctypes.c_float * len(STAGES))()
        cnt = (ctypes.c_uint32 * len(STAGES))()
        self._check(self._lib.vello_hip_get_stage_ms(self._h, ms, cnt), "get_stage_ms")
        return {STAGES[i]: (ms[i], cnt[i]) for i in range(len(STAGES))}

    # flatten: with one frame in flight the stroked lines' workgroups ride in the heavy list's launch (k_flatten_main) and what
    # they set aside follows (k_flatten_tail); with several, the stroked lines' kernel runs first and the heavy one takes it all
    FLATTEN_KERNELS = (("k_flatten_light", "k_flatten_main", "k_flatten_tail"), ("k_flatten_light", "k_flatten_strokes", "k_flatten_heavy"))
    KERNELS = {"flatten": FLATTEN_KERNELS[0], "coarse": ("k_coarse_prep", "k_coarse")}

    def kernel_ms(self):
        """vello_hip_get_kernel_ms for the stages that are several kernels: {kernel: (summed ms, profiled launches)}."""
        out = {}
        for stage, names in self.KERNELS.items():
            ms = (ctypes.c_float * 3)()
            cnt = ctypes.c_uint32()
            self._check(self._lib.vello_hip_get_kernel_ms(self._h, STAGES.index(stage), ms, ctypes.byref(cnt)), "get_kernel_ms")
            for k, n in enumerate(names):
                out[n] = (ms[k], cnt.value)
        return out


def estimate_capacities(packed, layout, width, height):
    """vello_hip_estimate_capacities: conservative pool sizes (dict) for a packed scene at a target size (host only)."""
    lib = load_library()
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    lay = LayoutStruct(*layout)
    p = RenderParamsStruct(width, height, 0, 0)
    c = Capacities()
    r = lib.vello_hip_estimate_capacities(packed.ctypes.data, packed.nbytes, ctypes.byref(lay), ctypes.byref(p), ctypes.byref(c))
    if r != 0:
        raise VelloHipError(f"vello_hip_estimate_capacities failed ({r})")
    return {k: getattr(c, k) for k, _ in Capacities._fields_}


def gather_frames(engines, src_frames, dst_frames, frame_bytes, dst_device=0, wait=True):
    """vello_hip_gather_frames / vello_hip_gather_wait: the frame each engine (one per GPU, one process) enqueued last is
    copied to dst_frames[i] on dst_device by SDMA peer copies (no CUs).  src / dst: torch tensors or device pointers."""
    lib = load_library()
    n = len(engines)

    def ptr(x):
        return ctypes.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))

    ctxs = (ctypes.c_void_p * n)(*[e._h for e in engines])
    srcs = (ctypes.c_void_p * n)(*[ptr(x) for x in src_frames])
    dsts = (ctypes.c_void_p * n)(*[ptr(x) for x in dst_frames])
    r = lib.vello_hip_gather_frames(ctxs, n, dst_device, srcs, dsts, frame_bytes)
    if r != 0:
        raise VelloHipError(f"vello_hip_gather_frames failed ({r})")
    if wait:
        r = lib.vello_hip_gather_wait(ctxs, n)
        if r != 0:
            raise VelloHipError(f"vello_hip_gather_wait failed ({r})")
