"""ctypes loader for libvello_hip.so (built in-tree by __graft_entry__.build / csrc/Makefile)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_PATH_OVERRIDE = None


class VelloHipError(RuntimeError):
    pass


def library_path():
    return _LIB_PATH_OVERRIDE or os.path.join(_HERE, "lib", "libvello_hip.so")


def _use_library(path):
    """Test hook: point the bindings at another build of the same sources (tests/simt_emu)."""
    global _LIB, _LIB_PATH_OVERRIDE
    _LIB = None
    _LIB_PATH_OVERRIDE = path


class Capacities(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("lines", "bin_data", "tiles", "seg_counts", "segments", "blend_spill", "ptcl")]


class Bump(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class LayoutStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "n_draw_objects", "n_paths", "n_clips", "bin_data_start", "path_tag_base", "path_data_base",
        "draw_tag_base", "draw_data_base", "transform_base", "style_base")]


class RenderParamsStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("width", "height", "base_color", "aa")]


def load_library():
    """Loads the product library; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise VelloHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). vello_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy is mapped first
    # serves the whole process, and torch cannot initialise on a foreign copy -> let torch map its runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    c = ctypes
    vp, u32, sz, i32 = c.c_void_p, c.c_uint32, c.c_size_t, c.c_int

    def sig(name, res, args):
        try:
            f = getattr(lib, name)
        except AttributeError:
            # an OLDER build of the library under the _use_library hook (same-box A/B against an earlier commit, scripts/ab_*.py)
            # may lack a later entry point; the product library must have them all
            if _LIB_PATH_OVERRIDE is not None and os.sep + "ab_tmp" + os.sep in _LIB_PATH_OVERRIDE:
                return
            raise
        f.restype = res
        f.argtypes = args

    # include/vello_hip.h
    sig("vello_hip_create", i32, [i32, u32, c.POINTER(Capacities), c.POINTER(vp)])
    sig("vello_hip_destroy", None, [vp])
    sig("vello_hip_render", i32, [vp, vp, sz, c.POINTER(LayoutStruct), c.POINTER(RenderParamsStruct), vp, u32, vp, sz, i32, c.POINTER(Bump)])
    sig("vello_hip_upload_scene", i32, [vp, vp, sz, c.POINTER(LayoutStruct), vp, u32])
    sig("vello_hip_render_frame", i32, [vp, vp, sz, c.POINTER(LayoutStruct), c.POINTER(RenderParamsStruct), vp, u32, vp, sz])
    sig("vello_hip_render_resident", i32, [vp, c.POINTER(RenderParamsStruct), vp, sz])
    sig("vello_hip_set_frames_in_flight", i32, [vp, u32])
    sig("vello_hip_resize_image_atlas", i32, [vp, u32, u32])
    sig("vello_hip_write_image", i32, [vp, u32, u32, u32, u32, vp, sz])
    sig("vello_hip_get_capacities", i32, [vp, c.POINTER(Capacities)])
    sig("vello_hip_grow_pools", i32, [vp, c.POINTER(Bump), c.POINTER(Capacities)])
    sig("vello_hip_set_auto_grow", i32, [vp, i32])
    sig("vello_hip_set_debug_flags", i32, [vp, u32])
    sig("vello_hip_last_render_attempts", u32, [vp])
    sig("vello_hip_fused_launches", ctypes.c_uint64, [vp])
    sig("vello_hip_estimate_capacities", i32, [vp, sz, c.POINTER(LayoutStruct), c.POINTER(RenderParamsStruct), c.POINTER(Capacities)])
    sig("vello_hip_gather_frames", i32, [c.POINTER(vp), u32, i32, c.POINTER(vp), c.POINTER(vp), sz])
    sig("vello_hip_gather_wait", i32, [c.POINTER(vp), u32])
    sig("vello_hip_sync_frame", i32, [vp, u32])
    sig("vello_hip_sync", i32, [vp])
    sig("vello_hip_get_bump", i32, [vp, c.POINTER(Bump)])
    sig("vello_hip_get_stream", vp, [vp])
    sig("vello_hip_run_stages", i32, [vp, c.POINTER(RenderParamsStruct), i32, i32])
    sig("vello_hip_read_buffer", i32, [vp, i32, vp, sz, sz])
    sig("vello_hip_write_buffer", i32, [vp, i32, vp, sz, sz])
    sig("vello_hip_buffer_size", sz, [vp, i32])
    sig("vello_hip_set_profiling", i32, [vp, u32])
    sig("vello_hip_get_stage_ms", i32, [vp, c.POINTER(c.c_float), c.POINTER(u32)])
    sig("vello_hip_get_kernel_ms", i32, [vp, i32, c.POINTER(c.c_float), c.POINTER(u32)])
    sig("vello_hip_stage_name", c.c_char_p, [i32])
    sig("vello_hip_last_error", c.c_char_p, [vp])
    sig("vello_hip_make_mask_lut", None, [vp])
    sig("vello_hip_make_mask_lut_16", None, [vp])
    # host mirror (harness glue)
    dp = c.POINTER(c.c_double)
    fp = c.POINTER(c.c_float)
    sig("vh_bezpath_free", None, [vp])
    sig("vh_bezpath_from_svg", vp, [c.c_char_p])
    sig("vh_bezpath_circle", vp, [c.c_double] * 4)
    sig("vh_bezpath_rect", vp, [c.c_double] * 4)
    sig("vh_bezpath_rounded_rect", vp, [c.c_double] * 6)
    sig("vh_bezpath_line", vp, [c.c_double] * 4)
    sig("vh_bezpath_n_verbs", sz, [vp])
    sig("vh_bezpath_n_coords", sz, [vp])
    sig("vh_bezpath_copy", None, [vp, vp, vp])
    sig("vh_scene_new", vp, [])
    sig("vh_scene_free", None, [vp])
    sig("vh_scene_reset", None, [vp])
    sig("vh_scene_fill", None, [vp, i32, dp, fp, vp, vp, sz])
    sig("vh_scene_stroke", i32, [vp, c.c_double, i32, c.c_double, i32, i32, dp, fp, vp, vp, sz])
    sig("vh_scene_push_layer", None, [vp, i32, u32, u32, c.c_float, dp, vp, vp, sz])
    sig("vh_scene_push_luminance_mask_layer", None, [vp, i32, c.c_float, dp, vp, vp, sz])
    sig("vh_scene_push_clip_layer", None, [vp, i32, dp, vp, vp, sz])
    sig("vh_scene_push_layer_stroked", i32, [vp, i32, c.c_double, i32, c.c_double, i32, i32, u32, u32, c.c_float, dp, vp, vp, sz])
    sig("vh_scene_pop_layer", None, [vp])
    sig("vh_scene_append", None, [vp, vp, dp])
    sig("vh_brush_solid", vp, [fp])
    sig("vh_brush_gradient", vp, [i32, dp, u32, u32, fp, sz])
    sig("vh_brush_image", vp, [c.c_uint64, u32, u32, u32, u32, vp, u32, u32, u32, c.c_float])
    sig("vh_brush_free", None, [vp])
    sig("vh_scene_fill_brush", None, [vp, i32, dp, vp, dp, vp, vp, sz])
    sig("vh_scene_stroke_brush", i32, [vp, c.c_double, i32, c.c_double, i32, i32, dp, vp, dp, vp, vp, sz])
    sig("vh_scene_draw_blurred_rounded_rect", None, [vp, dp, dp, fp, c.c_double, c.c_double])
    sig("vh_scene_draw_blurred_rounded_rect_in", None, [vp, vp, vp, sz, dp, dp, fp, c.c_double, c.c_double])
    sig("vh_scene_draw_image", None, [vp, vp, dp])
    sig("vh_scene_n_patches", sz, [vp])
    sig("vh_resolver_new", vp, [])
    sig("vh_resolver_new_with_atlas_sizes", vp, [u32, u32])
    sig("vh_resolver_free", None, [vp])
    sig("vh_resolver_resolve", sz, [vp, vp, c.POINTER(vp), c.POINTER(u32), c.POINTER(vp), c.POINTER(u32)])
    sig("vh_resolver_upload", vp, [vp, u32, c.POINTER(u32)])
    sig("vh_resolver_mark_image_dirty", None, [vp, c.c_uint64])
    sig("vh_resolver_image_cache_info", None, [vp, c.POINTER(u32)])
    sig("vh_scene_stream_bytes", sz, [vp, i32])
    sig("vh_scene_stream_copy", None, [vp, i32, vp])
    sig("vh_scene_counts", None, [vp, c.POINTER(u32)])
    sig("vh_scene_resolve", sz, [vp, c.POINTER(vp), c.POINTER(u32)])
    sig("vh_f32_to_f16", c.c_uint16, [c.c_float])
    sig("vh_f16_to_f32", c.c_float, [c.c_uint16])
    sig("vh_color_premul_rgba8", u32, [fp])
    sig("vh_style_from_stroke", None, [c.c_double, i32, c.c_double, i32, i32, c.POINTER(u32)])
    sig("vh_renderer_new", vp, [i32, u32, c.POINTER(Capacities), c.c_char_p, sz])
    sig("vh_renderer_free", None, [vp])
    sig("vh_renderer_render_to_texture", i32, [vp, vp, vp, sz, i32, u32, u32, fp, u32])
    sig("vh_renderer_error", c.c_char_p, [vp])
    sig("vh_renderer_engine", vp, [vp])
    sig("vh_renderer_last_bump", None, [vp, c.POINTER(Bump)])
    _LIB = lib
    return lib


def kernel_sources_hash():
    """sha1 over the kernel sources (vello_amd/csrc/engine/*, csrc/Makefile): what a profile or a PMC pass says it measured.  A
    commit hash moves with every documentation commit and does not exist on the GPU box; this says whether the KERNELS are the ones
    the numbers came from (bench.py: roofline.traffic_stale)."""
    import hashlib
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(here, "csrc")
    h = hashlib.sha1()
    # an explicit list: the Makefile, every source the kernel objects depend on -- engine/*.hip, *.h, *.inc and the
    # public header with the layout / config / flag types the kernels use -- and nothing else (no editor backups, no stray files)
    engine = os.path.join(root, "engine")
    files = [os.path.join(root, "Makefile"), os.path.join(os.path.dirname(here), "include", "vello_hip.h")]
    files += sorted(os.path.join(engine, f) for f in os.listdir(engine) if f.endswith((".hip", ".h", ".inc")))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())
            h.update(fh.read())
    return h.hexdigest()[:12]
