//! `extern "C"` mirror of include/vello_hip.h.  Kept in step with the header by tests/test_shim.py (it parses both
//! files and compares every function's name, arity and argument / return types, every struct's fields and every
//! constant).  Pointers to vello_encoding's `Layout` / `BumpAllocators` are passed as the header's own structs: both
//! are `#[repr(C)]` Pod with the same fields (vello_encoding/src/resolve.rs:18-39, config.rs:24-37).
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct vello_hip_ctx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct vello_hip_layout {
    pub n_draw_objects: u32,
    pub n_paths: u32,
    pub n_clips: u32,
    pub bin_data_start: u32,
    pub path_tag_base: u32,
    pub path_data_base: u32,
    pub draw_tag_base: u32,
    pub draw_data_base: u32,
    pub transform_base: u32,
    pub style_base: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct vello_hip_render_params {
    pub width: u32,
    pub height: u32,
    pub base_color: u32,
    pub aa: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct vello_hip_bump {
    pub failed: u32,
    pub binning: u32,
    pub ptcl: u32,
    pub tile: u32,
    pub seg_counts: u32,
    pub segments: u32,
    pub blend: u32,
    pub lines: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct vello_hip_capacities {
    pub lines: u32,
    pub bin_data: u32,
    pub tiles: u32,
    pub seg_counts: u32,
    pub segments: u32,
    pub blend_spill: u32,
    pub ptcl: u32,
}

pub const VELLO_HIP_AA_AREA: u32 = 0;
pub const VELLO_HIP_AA_MSAA8: u32 = 1;
pub const VELLO_HIP_AA_MSAA16: u32 = 2;
pub const VELLO_HIP_AA_MASK_AREA: u32 = 1;
pub const VELLO_HIP_AA_MASK_MSAA8: u32 = 2;
pub const VELLO_HIP_AA_MASK_MSAA16: u32 = 4;
pub const VELLO_HIP_AA_MASK_ALL: u32 = 7;
pub const VELLO_HIP_OK: c_int = 0;
pub const VELLO_HIP_E_INVALID: c_int = -1;
pub const VELLO_HIP_E_HIP: c_int = -2;
pub const VELLO_HIP_E_NO_DEVICE: c_int = -3;
pub const VELLO_HIP_E_CAPACITY: c_int = -4;
pub const VELLO_HIP_E_INTERNAL: c_int = -5;
pub const VELLO_HIP_DEBUG_NO_CULL: u32 = 1;
pub const VELLO_HIP_DEBUG_STROKE_KERNEL: u32 = 2;
pub const VELLO_HIP_DEBUG_SEQ_CLIP: u32 = 4;
pub const VELLO_HIP_DEBUG_FINE_SLICES: u32 = 8;
pub const VELLO_HIP_DEBUG_FLATTEN_COOP: u32 = 16;
pub const VELLO_HIP_DEBUG_FLATTEN_ALONE: u32 = 32;
pub const VELLO_HIP_DEBUG_NO_FUSION: u32 = 64;
pub const VELLO_HIP_STAGE_COUNT: usize = 11;

unsafe extern "C" {
    pub fn vello_hip_create(device: c_int, aa_mask: u32, caps: *const vello_hip_capacities, out: *mut *mut vello_hip_ctx) -> c_int;
    pub fn vello_hip_destroy(ctx: *mut vello_hip_ctx);
    pub fn vello_hip_render(ctx: *mut vello_hip_ctx, scene: *const u8, scene_len: usize, layout: *const vello_hip_layout, params: *const vello_hip_render_params, ramps: *const u32, n_ramps: u32, out_rgba8: *mut c_void, out_stride: usize, out_is_device: c_int, bump_out: *mut vello_hip_bump) -> c_int;
    pub fn vello_hip_upload_scene(ctx: *mut vello_hip_ctx, scene: *const u8, scene_len: usize, layout: *const vello_hip_layout, ramps: *const u32, n_ramps: u32) -> c_int;
    pub fn vello_hip_render_resident(ctx: *mut vello_hip_ctx, params: *const vello_hip_render_params, out_device: *mut c_void, out_stride: usize) -> c_int;
    pub fn vello_hip_render_frame(ctx: *mut vello_hip_ctx, scene: *const u8, scene_len: usize, layout: *const vello_hip_layout, params: *const vello_hip_render_params, ramps: *const u32, n_ramps: u32, out_device: *mut c_void, out_stride: usize) -> c_int;
    pub fn vello_hip_resize_image_atlas(ctx: *mut vello_hip_ctx, width: u32, height: u32) -> c_int;
    pub fn vello_hip_write_image(ctx: *mut vello_hip_ctx, x: u32, y: u32, width: u32, height: u32, rgba8: *const u8, stride: usize) -> c_int;
    pub fn vello_hip_get_capacities(ctx: *mut vello_hip_ctx, out: *mut vello_hip_capacities) -> c_int;
    pub fn vello_hip_grow_pools(ctx: *mut vello_hip_ctx, demand: *const vello_hip_bump, new_caps: *mut vello_hip_capacities) -> c_int;
    pub fn vello_hip_last_render_attempts(ctx: *mut vello_hip_ctx) -> u32;
    pub fn vello_hip_fused_launches(ctx: *mut vello_hip_ctx) -> u64;
    pub fn vello_hip_estimate_capacities(scene: *const u8, scene_len: usize, layout: *const vello_hip_layout, params: *const vello_hip_render_params, out: *mut vello_hip_capacities) -> c_int;
    pub fn vello_hip_set_auto_grow(ctx: *mut vello_hip_ctx, enabled: c_int) -> c_int;
    pub fn vello_hip_set_debug_flags(ctx: *mut vello_hip_ctx, flags: u32) -> c_int;
    pub fn vello_hip_set_frames_in_flight(ctx: *mut vello_hip_ctx, n: u32) -> c_int;
    pub fn vello_hip_sync_frame(ctx: *mut vello_hip_ctx, age: u32) -> c_int;
    pub fn vello_hip_sync(ctx: *mut vello_hip_ctx) -> c_int;
    pub fn vello_hip_get_bump(ctx: *mut vello_hip_ctx, out: *mut vello_hip_bump) -> c_int;
    pub fn vello_hip_get_stream(ctx: *mut vello_hip_ctx) -> *mut c_void;
    pub fn vello_hip_gather_frames(ctxs: *const *mut vello_hip_ctx, n: u32, dst_device: c_int, src_frames: *const *const c_void, dst_frames: *const *mut c_void, frame_bytes: usize) -> c_int;
    pub fn vello_hip_gather_wait(ctxs: *const *mut vello_hip_ctx, n: u32) -> c_int;
    pub fn vello_hip_run_stages(ctx: *mut vello_hip_ctx, params: *const vello_hip_render_params, first_stage: c_int, last_stage: c_int) -> c_int;
    pub fn vello_hip_read_buffer(ctx: *mut vello_hip_ctx, buf_id: c_int, dst: *mut c_void, offset: usize, size: usize) -> c_int;
    pub fn vello_hip_write_buffer(ctx: *mut vello_hip_ctx, buf_id: c_int, src: *const c_void, offset: usize, size: usize) -> c_int;
    pub fn vello_hip_buffer_size(ctx: *mut vello_hip_ctx, buf_id: c_int) -> usize;
    pub fn vello_hip_set_profiling(ctx: *mut vello_hip_ctx, stage_mask: u32) -> c_int;
    pub fn vello_hip_get_stage_ms(ctx: *mut vello_hip_ctx, ms_out: *mut f32, count_out: *mut u32) -> c_int;
    pub fn vello_hip_get_kernel_ms(ctx: *mut vello_hip_ctx, stage: c_int, ms_out: *mut f32, count_out: *mut u32) -> c_int;
    pub fn vello_hip_stage_name(stage: c_int) -> *const c_char;
    pub fn vello_hip_last_error(ctx: *mut vello_hip_ctx) -> *const c_char;
    pub fn vello_hip_make_mask_lut(out: *mut u8);
    pub fn vello_hip_make_mask_lut_16(out: *mut u8);
}
