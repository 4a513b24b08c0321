/*
 * vello_hip.h -- C ABI of the MI355X (gfx950) engine that replaces vello's wgpu compute path.
 *
 * Drop-in boundary: everything above this file stays the reference's host code
 * (vello::Scene -> vello_encoding::Encoding -> Resolver::resolve -> packed scene bytes + Layout);
 * everything below it (vello/src/wgpu_engine.rs WgpuEngine::run_recording, the Recording built by
 * vello/src/render.rs:135-629, and the 22 WGSL entry points of vello_shaders/shader/) is replaced
 * by this library.  Plain pointers and sizes only; no C++/torch types.
 *
 * Reference interface replaced by each entry point:
 *   vello_hip_create           vello::Renderer::new                      vello/src/lib.rs:432-459
 *                              (+ shaders::full_shaders                  vello/src/shaders.rs:48-274)
 *   vello_hip_destroy          Drop for Renderer / WgpuEngine
 *   vello_hip_render           Renderer::render_to_texture               vello/src/lib.rs:474-515
 *                              = render_full + WgpuEngine::run_recording vello/src/render.rs:84-112,
 *                                                                        vello/src/wgpu_engine.rs:380-777
 *   vello_hip_render_frame     render_to_texture without the wait        vello/src/lib.rs:474-515, wgpu_engine.rs:757
 *   vello_hip_upload_scene     Command::Upload("vello.scene") +          vello/src/render.rs:229-232,
 *                              Command::UploadUniform("vello.config")    vello/src/recording.rs:124-140
 *   vello_hip_render_resident  the Dispatch/DispatchIndirect chain       vello/src/render.rs:250-502, :560-629
 *   vello_hip_resize_image_atlas  ImageProxy::new(atlas_width, atlas_height)  vello/src/render.rs:160-176
 *   vello_hip_write_image      Recording::write_image(image_atlas, x, y, ..) vello/src/render.rs:201-203
 *   vello_hip_sync             queue.submit + device.poll                vello/src/wgpu_engine.rs:757
 *   vello_hip_set_frames_in_flight  back-to-back queue.submit without waiting  vello/src/wgpu_engine.rs:757
 *   vello_hip_get_bump         the robust path's bump download           vello/src/lib.rs:730, :753-761
 *   vello_hip_grow_pools /     "TODO: apply logic to determine whether   vello/src/lib.rs:762-764,
 *   vello_hip_set_auto_grow    we need to rerun coarse" + pool sizes     vello_encoding/src/config.rs:398-408
 *   vello_hip_estimate_capacities  BumpEstimator::count_path / tally      vello_encoding/src/estimate.rs:54-190
 *   vello_hip_gather_frames /  (none upstream: one Renderer per wgpu Device;  SURVEY.md 8e
 *   vello_hip_gather_wait      the scenes-per-GPU exchange of BASELINE config C5)
 *   vello_hip_set_debug_flags  (test seam: reference-exact coarse output)  vello_shaders/shader/coarse.wgsl:156-471
 *   vello_hip_run_stages /     CpuShaderType::Present per-stage seam     vello/src/wgpu_engine.rs:57-61, :541-553,
 *   vello_hip_{read,write}_buffer  (CpuBinding byte buffers)             vello_shaders/src/cpu.rs:58-62
 *   vello_hip_set_profiling /  wgpu-profiler per-dispatch GPU timestamps vello/src/wgpu_engine.rs:570-588
 *   vello_hip_get_stage_ms
 */
#ifndef VELLO_HIP_H
#define VELLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vello_hip_ctx vello_hip_ctx;

/* vello_encoding::Layout, vello_encoding/src/resolve.rs:18-39 (10 x u32, offsets in u32 words) */
typedef struct vello_hip_layout {
    uint32_t n_draw_objects, n_paths, n_clips, bin_data_start;
    uint32_t path_tag_base, path_data_base, draw_tag_base, draw_data_base;
    uint32_t transform_base, style_base;
} vello_hip_layout;

/* vello::AaConfig, vello/src/lib.rs:175-193 */
enum { VELLO_HIP_AA_AREA = 0, VELLO_HIP_AA_MSAA8 = 1, VELLO_HIP_AA_MSAA16 = 2 };
/* vello::AaSupport bits for vello_hip_create(aa_mask) */
enum { VELLO_HIP_AA_MASK_AREA = 1, VELLO_HIP_AA_MASK_MSAA8 = 2, VELLO_HIP_AA_MASK_MSAA16 = 4, VELLO_HIP_AA_MASK_ALL = 7 };

/* vello::RenderParams, vello/src/lib.rs:357-369.  base_color is premultiplied RGBA8 packed with
 * R in the low byte (vello_encoding/src/config.rs:183). */
typedef struct vello_hip_render_params {
    uint32_t width, height;
    uint32_t base_color;
    uint32_t aa; /* VELLO_HIP_AA_* */
} vello_hip_render_params;

/* vello_encoding::BumpAllocators, vello_encoding/src/config.rs:24-37 */
typedef struct vello_hip_bump {
    uint32_t failed, binning, ptcl, tile, seg_counts, segments, blend, lines;
} vello_hip_bump;

/* Capacities of the bump-allocated pools in elements.  Zero fields take the reference's
 * hand-picked sizes (vello_encoding/src/config.rs:398-408). */
typedef struct vello_hip_capacities {
    uint32_t lines, bin_data, tiles, seg_counts, segments, blend_spill, ptcl;
} vello_hip_capacities;

/* Error codes (0 = ok).  Capacity overflow mirrors the reference protocol: the target is left
 * untouched (fine.wgsl:1070-1074) and additionally VELLO_HIP_E_CAPACITY is returned by
 * vello_hip_sync / vello_hip_render with the counters available through vello_hip_get_bump. */
enum {
    VELLO_HIP_OK = 0,
    VELLO_HIP_E_INVALID = -1,   /* bad argument / AA mode not enabled at create (render.rs:566-598 panics) / a packed scene
                                 * whose streams contradict each other: draw tags that need more draw data, info words,
                                 * clips or paths than the layout provides (refused at upload), or path tags that need more
                                 * path data, transforms or styles than the scene holds (found by the pathtag scan; the
                                 * target is left untouched and vello_hip_render / vello_hip_sync report it).  WebGPU's
                                 * robust buffer access absorbs such scenes upstream; HIP has none, so they are refused. */
    VELLO_HIP_E_HIP = -2,       /* HIP runtime error; see vello_hip_last_error */
    VELLO_HIP_E_NO_DEVICE = -3, /* no gfx950 device / kernels missing: never falls back to a CPU path */
    VELLO_HIP_E_CAPACITY = -4,  /* bump.failed != 0: a pool overflowed (the robust path grows the pools and renders again) */
    VELLO_HIP_E_INTERNAL = -5   /* a spin bound of the engine tripped (a look-back or k_front's grid barrier waited for a workgroup
                                 * that never arrived): the frame is discarded, the lane's counters are reset; NOT a pool overflow --
                                 * auto-grow does not treat it as one */
};

/* Stage ids (launch order; render.rs:250-502, :560-629).  Several reference dispatches are fused:
 * PATHTAG_SCAN = pathtag_reduce(+2)/scan(1)/scan + bbox_clear, DRAW_SCAN = draw_reduce + draw_leaf,
 * CLIP = clip_reduce + clip_leaf, PATH_COUNT/PATH_TILING include their *_setup dispatch.
 * A frame (and any vello_hip_run_stages range that holds both FLATTEN and DRAW_SCAN) runs DRAW_SCAN's workgroups in
 * FLATTEN's first launch -- the stage needs the scene and the pathtag scan only -- and then has no launch of its own: its
 * profiled time is an empty event pair; a range that starts at DRAW_SCAN launches it as a kernel. */
enum {
    VELLO_HIP_STAGE_PATHTAG_SCAN = 0,
    VELLO_HIP_STAGE_FLATTEN,
    VELLO_HIP_STAGE_DRAW_SCAN,
    VELLO_HIP_STAGE_CLIP,
    VELLO_HIP_STAGE_BINNING,
    VELLO_HIP_STAGE_TILE_ALLOC,
    VELLO_HIP_STAGE_PATH_COUNT,
    VELLO_HIP_STAGE_BACKDROP,
    VELLO_HIP_STAGE_COARSE,
    VELLO_HIP_STAGE_PATH_TILING,
    VELLO_HIP_STAGE_FINE,
    VELLO_HIP_STAGE_COUNT
};

/* Buffer ids for the differential-test seam (byte layouts = the reference's). */
enum {
    VELLO_HIP_BUF_SCENE = 0,
    VELLO_HIP_BUF_CONFIG,        /* ConfigUniform, 88 B */
    VELLO_HIP_BUF_TAG_MONOIDS,   /* PathMonoid[Tw], 20 B */
    VELLO_HIP_BUF_PATH_BBOXES,   /* PathBbox[P], 24 B */
    VELLO_HIP_BUF_BUMP,          /* BumpAllocators, 32 B */
    VELLO_HIP_BUF_LINES,         /* LineSoup[], 24 B */
    VELLO_HIP_BUF_DRAW_MONOIDS,  /* DrawMonoid[D], 16 B */
    VELLO_HIP_BUF_INFO_BIN_DATA, /* u32[] */
    VELLO_HIP_BUF_CLIP_INP,      /* Clip[K], 8 B */
    VELLO_HIP_BUF_CLIP_BBOXES,   /* f32x4[K] */
    VELLO_HIP_BUF_DRAW_BBOXES,   /* f32x4[D] */
    VELLO_HIP_BUF_BIN_HEADERS,   /* BinHeader[], 8 B */
    VELLO_HIP_BUF_PATHS,         /* Path[], 32 B */
    VELLO_HIP_BUF_TILES,         /* Tile[], 8 B */
    VELLO_HIP_BUF_SEG_COUNTS,    /* SegmentCount[], 8 B */
    VELLO_HIP_BUF_SEGMENTS,      /* PathSegment[], 24 B */
    VELLO_HIP_BUF_PTCL,          /* u32[] */
    VELLO_HIP_BUF_BLEND_SPILL,   /* u32[] */
    VELLO_HIP_BUF_OUTPUT,        /* internal RGBA8 target, width*height*4 */
    VELLO_HIP_BUF_COUNT
};

/* Creates an engine bound to HIP device `device`.  Fails with VELLO_HIP_E_NO_DEVICE when no GPU is
 * present; there is no CPU fallback. */
int vello_hip_create(int device, uint32_t aa_mask, const vello_hip_capacities *caps /* nullable */, vello_hip_ctx **out);
void vello_hip_destroy(vello_hip_ctx *ctx);

/* One frame, host buffers in, blocking: upload + render + (optional) copy out.
 * `out_rgba8` receives un-premultiplied RGBA8 rows of `out_stride` bytes (fine.wgsl:1386-1397);
 * it is a device pointer when out_is_device != 0, else host memory.  `ramps` is the gradient
 * ramp texture (512 RGBA8 texels per ramp, vello_encoding/src/ramp_cache.rs:12) or NULL.
 * `bump_out` (nullable) receives the bump counters. */
int vello_hip_render(vello_hip_ctx *ctx, const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                     const vello_hip_render_params *params, const uint32_t *ramps, uint32_t n_ramps, void *out_rgba8,
                     size_t out_stride, int out_is_device, vello_hip_bump *bump_out);

/* Split form used for steady-state measurement: the packed scene is made resident once ... */
int vello_hip_upload_scene(vello_hip_ctx *ctx, const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                           const uint32_t *ramps, uint32_t n_ramps);
/* ... then each call enqueues one full frame (all stages) on the context's stream and returns
 * without waiting.  `out_device` may be NULL (render into the internal target only).  Resident frames ALWAYS show
 * the scene of the last vello_hip_upload_scene: scenes passed to vello_hip_render_frame are private to their frame
 * (VELLO_HIP_E_INVALID if no scene was ever uploaded).  The frame runs on the context's own (non-blocking) stream
 * (vello_hip_get_stream): work of the caller's streams on `out_device` -- clearing it, reading the previous frame -- has
 * to be finished or ordered against that stream by the caller, as with any wgpu texture shared between queues. */
int vello_hip_render_resident(vello_hip_ctx *ctx, const vello_hip_render_params *params, void *out_device, size_t out_stride);
/* Animation form (every frame has its own scene): vello_hip_upload_scene + vello_hip_render_resident in one call
 * that does NOT wait for the frame.  The scene is copied into the private slot of the next in-flight buffer set
 * (only that set's previous frame is waited for), so with vello_hip_set_frames_in_flight(n > 1) the upload of frame
 * i+1 overlaps the rendering of frames i, i-1, ...  `scene` / `ramps` may be reused when the call returns; the target
 * is complete after vello_hip_sync_frame(0) / vello_hip_sync.  Replaces, per frame, the same reference calls as
 * vello_hip_render (vello/src/lib.rs:474-515) under wgpu's submit-without-wait (vello/src/wgpu_engine.rs:757). */
int vello_hip_render_frame(vello_hip_ctx *ctx, const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                           const vello_hip_render_params *params, const uint32_t *ramps, uint32_t n_ramps, void *out_device,
                           size_t out_stride);

/* The image atlas: one RGBA8 texture that persists across frames (render.rs:160-176).  The Resolver owns
 * the packing: it patches every DrawImage's atlas xy (resolve.rs:300-316) and lists the images to (re)write.
 * resize discards the contents (zero-filled), as creating a new ImageProxy does.  Texel bytes are stored
 * verbatim; BGRA / straight-alpha images are normalised while sampling (fine.wgsl:829-858).
 * resize waits for the frames in flight.  write_image does not (wgpu's queue.write_texture is queued as well): it copies
 * the caller's pixels to pinned memory during the call and enqueues the transfer in submission order -- behind every
 * frame enqueued before it (they still sample the old texels), in front of every frame enqueued after it. */
int vello_hip_resize_image_atlas(vello_hip_ctx *ctx, uint32_t width, uint32_t height);
int vello_hip_write_image(vello_hip_ctx *ctx, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const uint8_t *rgba8,
                          size_t stride /* bytes per source row; 0 = width*4 */);

/* Robust dynamic memory.  vello_hip_grow_pools re-sizes every pool whose counter in `demand`
 * (from vello_hip_get_bump / bump_out after VELLO_HIP_E_CAPACITY) exceeds it, with 25 % headroom, on all in-flight
 * buffer sets; the caller then renders the frame again.  A stage that overflows stops the later stages
 * (shared/bump.wgsl:5-9), so one frame may need several rounds.  Returns VELLO_HIP_E_INVALID when nothing had to
 * grow.  With vello_hip_set_auto_grow(ctx, 1) the blocking vello_hip_render does these rounds itself and only
 * reports VELLO_HIP_E_CAPACITY if the demand cannot be met; every entry point that renders then also sizes the PTCL
 * pool for the target (64 words per tile are fixed, config.rs:408 allows ~2 Mpx of tiles) instead of failing with
 * VELLO_HIP_E_INVALID. */
int vello_hip_get_capacities(vello_hip_ctx *ctx, vello_hip_capacities *out);
int vello_hip_grow_pools(vello_hip_ctx *ctx, const vello_hip_bump *demand, vello_hip_capacities *new_caps /* nullable */);
int vello_hip_set_auto_grow(vello_hip_ctx *ctx, int enabled);
/* vello_encoding::BumpEstimator (vello_encoding/src/estimate.rs:54-190) applied to the packed scene: conservative
 * pool sizes for this scene at this target size, computed on the host without touching the GPU.  Lines and segments
 * follow the reference's counting rules (Wang's formula for curves, sqrt(2)-inflated tile crossings, arc counts of
 * round joins / caps); tiles, bin entries and PTCL words -- TODO upstream (estimate.rs:14-16) -- come from the paths'
 * control-point bounding boxes.  blend_spill is not estimated (0).  With vello_hip_set_auto_grow the blocking
 * vello_hip_render calls this itself when the scene is large against the current pools. */
/* How many rounds the last blocking vello_hip_render took (1 unless robust mode had to grow pools and re-run). */
uint32_t vello_hip_last_render_attempts(vello_hip_ctx *ctx);
/* Launches in which the stages of a small scene shared a kernel (VELLO_HIP_DEBUG_NO_FUSION), counted since the context was
 * created: lets a test see that the scene it renders took that path. */
uint64_t vello_hip_fused_launches(vello_hip_ctx *ctx);
int vello_hip_estimate_capacities(const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                                  const vello_hip_render_params *params, vello_hip_capacities *out);

/* Test-seam switches (default 0).  VELLO_HIP_DEBUG_NO_CULL turns off coarse's occlusion culling (a draw hidden under a
 * later opaque full-tile cover is normally not emitted; the image is the same, but bump.segments / bump.ptcl and the
 * PTCL words are then <= the reference's): with it set, PTCL, segment slices and every bump counter equal the
 * reference's (coarse.wgsl:156-471) up to the order its atomics hand out slices and chunks.
 * VELLO_HIP_DEBUG_STROKE_KERNEL runs the stroked-line kernel of flatten for any number of stroked lines (it normally takes
 * over from 393 216 of them): same lines, a different kernel -- so that small test scenes exercise it.
 * VELLO_HIP_DEBUG_SEQ_CLIP matches clips with the one-wave stack machine that otherwise only takes scenes of more than
 * 524 288 clips (clip_reduce.wgsl / clip_leaf.wgsl run as partitioned kernels below that): same clip boxes.
 * VELLO_HIP_DEBUG_FINE_SLICES cuts EVERY tile's command list into slices of 4 fills for fine's MSAA modes (normally only
 * lists of >= 96 fills are cut, into slices of 32: the slices' coverage is computed by separate waves and the last one to
 * finish composites the tile): same image -- so that small test scenes exercise the sliced path.
 * VELLO_HIP_DEBUG_FLATTEN_COOP / _ALONE pick the kernels that flatten the scene's curves, stroked curves, joins and caps: the
 * wave-cooperative walk, or every lane on its own (normally the engine picks by what an earlier frame of the same scene put on
 * the list, and by the scene's size before there is one): same line soup as a multiset -- so that tests can hold both sets of
 * kernels to the oracle on the same scenes.
 * VELLO_HIP_DEBUG_NO_FUSION launches every stage of a small scene as a kernel of its own (normally the workgroups of consecutive
 * stages up to tile_alloc share launches when the scene is small enough for launch boundaries to matter): same buffers. */
enum { VELLO_HIP_DEBUG_NO_CULL = 1, VELLO_HIP_DEBUG_STROKE_KERNEL = 2, VELLO_HIP_DEBUG_SEQ_CLIP = 4, VELLO_HIP_DEBUG_FINE_SLICES = 8,
       VELLO_HIP_DEBUG_FLATTEN_COOP = 16, VELLO_HIP_DEBUG_FLATTEN_ALONE = 32, VELLO_HIP_DEBUG_NO_FUSION = 64,
       /* measurement seam: bits 24-27 = 1 + the last stage vello_hip_render_resident launches (0: all of them) -- what the stages
        * up to k cost with frames in flight (scripts/experiments/r6_stage_marginal.py); the frames are incomplete */
       VELLO_HIP_DEBUG_LAST_STAGE_SHIFT = 24,
       /* experiment: bit (8 + stage) makes that stage run for one frame at a time when frames are in flight (its launches wait for the
        * same stage of the frame enqueued before) -- scripts/experiments/r6_exclusive_stages.py */
       VELLO_HIP_DEBUG_EXCLUSIVE_SHIFT = 8 };
int vello_hip_set_debug_flags(vello_hip_ctx *ctx, uint32_t flags);

/* Number of frames the context keeps in flight (default 1, max 8).  wgpu queues recordings without waiting
 * (wgpu_engine.rs:757); with n > 1 consecutive vello_hip_render_resident calls rotate over n private buffer
 * sets and streams and overlap on the GPU.  The caller must hand each in-flight frame its own target. */
int vello_hip_set_frames_in_flight(vello_hip_ctx *ctx, uint32_t n);
/* Waits for the frame enqueued `age` vello_hip_render_resident calls ago (0 = newest, age < frames in flight)
 * without draining the younger ones; does not inspect bump.failed (vello_hip_sync does). */
int vello_hip_sync_frame(vello_hip_ctx *ctx, uint32_t age);
/* Waits for everything enqueued; returns VELLO_HIP_E_CAPACITY if a frame overflowed. */
int vello_hip_sync(vello_hip_ctx *ctx);
int vello_hip_get_bump(vello_hip_ctx *ctx, vello_hip_bump *out);
/* The hipStream_t the context launches on (for callers that record their own events). */
void *vello_hip_get_stream(vello_hip_ctx *ctx);

/* Multi-GPU exchange for a host that owns one context per GPU in one process: the frame each context
 * enqueued last (src_frames[i], device memory of ctxs[i]'s GPU) is copied to dst_frames[i] on `dst_device` with
 * hipMemcpyPeerAsync on a per-context copy stream, ordered behind that frame by an event: SDMA over the peer's own xGMI
 * link, no CUs, all peers concurrently.  Returns without waiting; vello_hip_gather_wait blocks until the copies have
 * landed.  (One process per GPU gathers with RCCL instead: vello_amd/distributed.py, bench.py --gpus N.) */
int vello_hip_gather_frames(vello_hip_ctx *const *ctxs, uint32_t n, int dst_device, const void *const *src_frames, void *const *dst_frames,
                            size_t frame_bytes);
int vello_hip_gather_wait(vello_hip_ctx *const *ctxs, uint32_t n);

/* Differential-test seam: run stages [first, last] of the resident scene; read/write any buffer. */
int vello_hip_run_stages(vello_hip_ctx *ctx, const vello_hip_render_params *params, int first_stage, int last_stage);
int vello_hip_read_buffer(vello_hip_ctx *ctx, int buf_id, void *dst, size_t offset, size_t size);
int vello_hip_write_buffer(vello_hip_ctx *ctx, int buf_id, const void *src, size_t offset, size_t size);
size_t vello_hip_buffer_size(vello_hip_ctx *ctx, int buf_id);

/* Per-stage GPU timing with hipEvents on the launch stream.  stage_mask bit i enables events
 * around stage i; times accumulate until read.  vello_hip_get_stage_ms syncs, writes the summed
 * milliseconds and launch counts per stage, and resets the accumulators. */
int vello_hip_set_profiling(vello_hip_ctx *ctx, uint32_t stage_mask);
int vello_hip_get_stage_ms(vello_hip_ctx *ctx, float ms_out[VELLO_HIP_STAGE_COUNT], uint32_t count_out[VELLO_HIP_STAGE_COUNT]);
/* The stages that are several kernels, kernel by kernel (events between the launches while the stage is profiled):
 * VELLO_HIP_STAGE_FLATTEN -> k_flatten_light, k_flatten_main, k_flatten_tail with one frame in flight (k_flatten_light,
 * k_flatten_strokes, k_flatten_heavy with several: vello_hip_set_frames_in_flight); VELLO_HIP_STAGE_COARSE -> k_coarse_prep,
 * k_coarse (third entry 0).  Summed milliseconds since the last call and the number of profiled launches of the stage;
 * zeros for a stage of one kernel (its time is vello_hip_get_stage_ms's).  Reading resets the sums. */
int vello_hip_get_kernel_ms(vello_hip_ctx *ctx, int stage, float ms_out[3], uint32_t *count_out);

const char *vello_hip_stage_name(int stage);
const char *vello_hip_last_error(vello_hip_ctx *ctx /* nullable: last create() error */);

/* vello_encoding::make_mask_lut / make_mask_lut_16 (vello_encoding/src/mask.rs:36-98); exported so
 * the host can check the persistent LUT the engine keeps on the device. */
void vello_hip_make_mask_lut(uint8_t out[1024]);
void vello_hip_make_mask_lut_16(uint8_t out[8192]);

#ifdef __cplusplus
}
#endif
#endif
