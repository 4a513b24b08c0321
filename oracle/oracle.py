"""ctypes wrapper of the CPU oracle (oracle/libvello_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by vello_amd.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STAGES = ["pathtag_scan", "flatten", "draw_scan", "clip", "binning", "tile_alloc", "path_count", "backdrop", "coarse",
          "path_tiling", "fine"]
BUFFERS = ["tag_monoids", "path_bboxes", "bump", "lines", "draw_monoids", "info_bin_data", "clip_inp", "clip_bboxes",
           "draw_bboxes", "bin_headers", "paths", "tiles", "seg_counts", "segments", "ptcl", "blend_spill", "output"]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvello_oracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        vp, u32, i32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_size_t
        lib.vo_create.restype = vp
        lib.vo_create.argtypes = [u32]
        lib.vo_destroy.argtypes = [vp]
        lib.vo_set_scene.restype = i32
        lib.vo_set_scene.argtypes = [vp, vp, sz, vp, u32, u32, u32, i32]
        lib.vo_set_ramps.argtypes = [vp, vp, u32]
        lib.vo_set_image_atlas.argtypes = [vp, vp, u32, u32]
        lib.vo_get_config.restype = vp
        lib.vo_get_config.argtypes = [vp]
        lib.vo_run.restype = i32
        lib.vo_run.argtypes = [vp, i32, i32]
        lib.vo_render.restype = i32
        lib.vo_render.argtypes = [vp, vp]
        lib.vo_buffer.restype = vp
        lib.vo_buffer.argtypes = [vp, i32, ctypes.POINTER(sz)]
        lib.vo_make_mask_lut.argtypes = [vp]
        lib.vo_make_mask_lut_16.argtypes = [vp]
        lib.vo_set_threads.argtypes = [vp, i32]
        lib.vo_set_capacity_scale.argtypes = [vp, u32]
        lib.vo_get_capacity_scale.restype = u32
        lib.vo_get_capacity_scale.argtypes = [vp]
        _LIB = lib
    return _LIB


def make_mask_lut():
    out = np.zeros(1024, dtype=np.uint8)
    _lib().vo_make_mask_lut(out.ctypes.data)
    return out


def make_mask_lut_16():
    out = np.zeros(8192, dtype=np.uint8)
    _lib().vo_make_mask_lut_16(out.ctypes.data)
    return out


class Oracle:
    """capacity_scale: pools = scale x the reference's fixed sizes (config.rs:398-408).  auto_grow: a frame that overflows a pool
    (bump.failed's stage bits) doubles the scale -- up to max_capacity_scale -- and runs again from the first stage, so that
    scenes beyond any fixed pool (the fuzzer's extreme-value mode) are still checked against something."""

    CAPACITY_FAILURES = 0x1F  # STAGE_BINNING | TILE_ALLOC | FLATTEN | PATH_COUNT | COARSE (config.rs:39-44)

    def __init__(self, capacity_scale=1, auto_grow=False, max_capacity_scale=64):
        self._lib = _lib()
        self._h = self._lib.vo_create(capacity_scale)
        self.width = self.height = 0
        self.auto_grow = auto_grow
        self.max_capacity_scale = max_capacity_scale
        self.grown = 0  # times the pools were doubled
        self._scene_args = None

    def __del__(self):
        try:
            if self._h:
                self._lib.vo_destroy(self._h)
        except Exception:
            pass
        self._h = None

    def set_threads(self, n):
        self._lib.vo_set_threads(self._h, n)

    def set_scene(self, packed, layout, width, height, base_color_rgba8, aa):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        lay = (ctypes.c_uint32 * 10)(*layout)
        r = self._lib.vo_set_scene(self._h, packed.ctypes.data, packed.nbytes, lay, width, height, int(base_color_rgba8), int(aa))
        if r != 0:
            raise RuntimeError("vo_set_scene failed")
        self.width, self.height = width, height
        self._scene_args = (packed, tuple(layout), width, height, int(base_color_rgba8), int(aa))

    def capacity_scale(self):
        return int(self._lib.vo_get_capacity_scale(self._h))

    def _grow_if_overflowed(self):
        """True if the last run overflowed a pool and the pools were doubled (the caller runs again from the first stage)."""
        if not self.auto_grow or self._scene_args is None:
            return False
        if (self.bump()["failed"] & self.CAPACITY_FAILURES) == 0:
            return False
        scale = self.capacity_scale()
        if scale * 2 > self.max_capacity_scale:
            return False
        self._lib.vo_set_capacity_scale(self._h, scale * 2)
        self.set_scene(*self._scene_args)
        self.grown += 1
        return True

    def set_ramps(self, ramps):
        """Gradient ramp texture: n_ramps x 512 RGBA8 texels as uint32 (ramp_cache.rs), or None."""
        if ramps is None or len(ramps) == 0:
            self._lib.vo_set_ramps(self._h, None, 0)
            return
        ramps = np.ascontiguousarray(ramps, dtype=np.uint32)
        self._lib.vo_set_ramps(self._h, ramps.ctypes.data, ramps.size // 512)

    def set_image_atlas(self, atlas):
        """Image atlas as an HxWx4 uint8 array (render.rs:160-203), or None."""
        if atlas is None:
            self._lib.vo_set_image_atlas(self._h, None, 0, 0)
            return
        atlas = np.ascontiguousarray(atlas, dtype=np.uint8)
        self._lib.vo_set_image_atlas(self._h, atlas.ctypes.data, atlas.shape[1], atlas.shape[0])

    def config(self):
        p = self._lib.vo_get_config(self._h)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32)), shape=(22,)).copy()

    def run(self, first, last):
        first = STAGES.index(first) if isinstance(first, str) else first
        last = STAGES.index(last) if isinstance(last, str) else last
        while True:
            if self._lib.vo_run(self._h, first, last) != 0:
                raise RuntimeError("vo_run failed")
            if first != 0 or not self._grow_if_overflowed():  # (a range from the first stage can simply be run again)
                break

    def render(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        while True:
            r = self._lib.vo_render(self._h, out.ctypes.data)
            if r < 0:
                raise RuntimeError("vo_render failed")
            if not self._grow_if_overflowed():
                break
        return out

    def buffer(self, name, dtype=np.uint8):
        """Live view of an intermediate buffer (valid until the next set_scene)."""
        size = ctypes.c_size_t()
        p = self._lib.vo_buffer(self._h, BUFFERS.index(name), ctypes.byref(size))
        a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(size.value,))
        n = size.value - size.value % np.dtype(dtype).itemsize
        return a[:n].view(dtype)

    def bump(self):
        b = self.buffer("bump", np.uint32)[:8]
        return dict(zip(["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"], [int(v) for v in b]))
