// Host driver of the gfx950 engine and the C ABI of include/vello_hip.h.
// Replaces WgpuEngine::run_recording (vello/src/wgpu_engine.rs:380-777) and the Recording built by
// Render::render_encoding_coarse / record_fine (vello/src/render.rs:135-629): the recording is a
// fixed launch sequence here, the ResourcePool (wgpu_engine.rs:972-1000) becomes a set of device
// buffers owned by the context and reused across frames.
#include "../../../include/vello_hip.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

using namespace vk;

namespace {

std::string g_create_error;

const char *kStageNames[VELLO_HIP_STAGE_COUNT] = {"pathtag_scan", "flatten",  "draw_scan", "clip",        "binning", "tile_alloc",
                                                  "path_count",   "backdrop", "coarse",    "path_tiling", "fine"};

struct DevBuf {
    void *ptr = nullptr;
    size_t size = 0;
};

uint32_t align_up(uint32_t len, uint32_t alignment) { return len + ((0u - len) & (alignment - 1u)); }

}  // namespace

// One frame in flight = one lane: its own stream and its own set of transient buffers.  The packed scene,
// ramps and mask LUTs are shared read-only.  wgpu queues recordings without waiting (wgpu_engine.rs:757); here
// consecutive frames additionally overlap on the GPU (coarse launches only one workgroup per bin, fine and
// flatten are latency-bound, so a second frame fills the idle CUs): vello_hip_set_frames_in_flight.
// A packed scene made resident: the bytes, the ramp texture and everything the host derives from the Layout.
struct SceneSlot {
    DevBuf scene, ramps;
    vello_hip_layout layout{};
    size_t scene_len = 0;
    uint32_t n_tag_words = 0, n_pathtag_parts = 0, n_draw_parts = 0, n_ramps = 0;
    size_t zero_bytes = 0;
    bool brushes = false;   // gradient / image / blurred-rect draw objects present (selects fine's specialisation)
    bool resident = false;
    // stroked-line tags of the scene as k_flatten_light counted them in an earlier frame (-1: not known yet).  A property of
    // the scene alone; lets the host leave out stroke workgroups that would exit at once.
    int64_t stroke_lines = -1;
    // lines in the soup of a finished frame of this scene (-1 unknown): picks path_count's chunk size (path.hip, k_path_count<LPT>)
    int64_t soup_lines = -1;
    int64_t slice_demand = -1;  // slice items coarse asked for in a finished MSAA frame of this scene (max seen), -1 unknown
    // what k_flatten_light put on flatten's heavy list in a finished frame of this scene: fills' curves, stroked curves (+ the cap
    // markers of open subpaths), stroked lines (-1: not known yet): picks the kernels that take the list (Frame::flatten_coop)
    int64_t heavy_curves = -1, heavy_strokes = -1;
    uint64_t generation = 0;  // bumped by every upload into the slot: a lane's finished frame speaks for the scene it rendered only
};

struct Lane {
    hipStream_t stream = nullptr;
    SceneSlot own;          // vello_hip_render_frame: the scene of the frame this lane is rendering
    bool use_own = false;   // else the context's shared scene (vello_hip_upload_scene)
    DevBuf buf[VELLO_HIP_BUF_COUNT];  // SCENE / CONFIG entries unused (shared, see ctx)
    DevBuf zero_region;               // Control + look-back states (BUF_BUMP aliases its head)
    DevBuf clip_stack;
    DevBuf coarse_el;                 // coarse: CoarseEl per draw object
    DevBuf tile_bits;                 // coarse: 3 bits per tile of the pool, a word per 8 tiles
    DevBuf tile_order;                // coarse -> fine: tiles bucketed by command-list length
    DevBuf slice_items, slice_counters, cov;  // coarse -> fine: slices of long tiles, their arrival counters, coverage scratch
    DevBuf heavy_list;                // flatten: tag indices for the heavy code, 4 lists (one u32 per tag each, worst case)
    DevBuf arc_items;                 // flatten: arcs the stroke workgroups leave to the heavy code (64 B per segment, worst case)
    DevBuf front_sync;                // k_front's grid-barrier counter (zeroed once, when allocated)
    uint32_t front_sync_value = 0;    // ... and its value once every launch enqueued so far has run
    struct EvPair {
        int stage;
        hipEvent_t a, b;
        hipEvent_t mid[2];  // behind the stage's first / second kernel when it has more than one (flatten: 3, coarse: 2)
    };
    std::vector<EvPair> events;
    bool used = false;
    bool slices_on = false;   // the lane's latest frame ran with fine's slices enabled (an MSAA frame)
    uint32_t slice_cap_coarse = 0;  // slice blocks the lane's latest COARSE launch was told of (a later FINE must launch as many)
    uint32_t slice_fills_coarse = 0;  // ... and the slice size it cut with (0: no slices -- an area-AA frame)
    bool flatten_ran = false;  // the control block holds flatten's counts (a partial vello_hip_run_stages range may stop before it)
    uint64_t frame_generation = 0;  // slot_of(...).generation when the lane's latest frame was set up
    uint64_t atlas_epoch_seen = 0;  // ctx::atlas_epoch the lane's stream has been ordered behind
};

// Pinned staging block of one vello_hip_write_image: the caller's pixels are copied here during the call, the DMA into the
// atlas runs from it on the upload stream; free again once `done` has passed.
struct Staging {
    void *host = nullptr;
    size_t size = 0;
    hipEvent_t done = nullptr;
    bool busy = false;
};

struct vello_hip_ctx {
    int device = 0;
    uint32_t aa_mask = 0;
    vello_hip_capacities caps{};
    DevBuf config;
    DevBuf mask8, mask16;
    SceneSlot shared;  // vello_hip_upload_scene: one scene for every lane
    DevBuf atlas;  // persistent image atlas (render.rs:160-176), shared by all lanes
    uint32_t atlas_w = 0, atlas_h = 0;
    std::vector<Lane> lanes;
    uint32_t n_active = 1;  // lanes in the rotation (<= lanes.size(): shrinking keeps the buffers)
    uint32_t next_lane = 0, last_lane = 0;
    bool auto_grow = false;
    hipStream_t copy_stream = nullptr;  // vello_hip_gather_frames: this context's peer copy
    hipEvent_t frame_done = nullptr;
    // vello_hip_write_image: atlas uploads are stream-ordered, not host-synchronous (wgpu's queue.write_texture is queued
    // too, render.rs:160-203).  An upload waits for the frames enqueued before it (they may sample the texels it replaces)
    // and every frame enqueued after it waits for `atlas_ready`.
    hipStream_t upload_stream = nullptr;
    hipEvent_t atlas_ready = nullptr, lane_mark = nullptr;
    // experiment (round 6, VELLO_HIP_DEBUG_EXCLUSIVE_SHIFT): per stage, an event behind the stage's launches of the frame enqueued last
    hipEvent_t stage_turn[VELLO_HIP_STAGE_COUNT] = {};
    uint64_t atlas_epoch = 0;  // uploads enqueued so far
    std::vector<Staging> staging;
    uint32_t debug_flags = 0;  // VELLO_HIP_DEBUG_*
    bool force_brushes = false;  // pre-warm: run fine's brush specialisation on a scene without brushes
    uint32_t last_render_attempts = 0;  // rounds the last vello_hip_render needed (robust mode)
    uint64_t fused_launches = 0;  // k_front launches so far (vello_hip_fused_launches)
    // last frame
    Config cfg{};
    bool have_cfg = false;
    // profiling
    uint32_t prof_mask = 0;
    std::vector<hipEvent_t> event_pool;
    float stage_ms[VELLO_HIP_STAGE_COUNT] = {};
    uint32_t stage_count[VELLO_HIP_STAGE_COUNT] = {};
    float kernel_ms[VELLO_HIP_STAGE_COUNT][3] = {};  // per kernel of the stages that are several (flatten, coarse)
    uint32_t kernel_count[VELLO_HIP_STAGE_COUNT] = {};
    std::string last_error;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                             \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return VELLO_HIP_E_HIP;                                                                    \
        }                                                                                              \
    } while (0)

int ensure(vello_hip_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 256;
    if (b.size >= bytes) return 0;
    if (b.ptr) HIP_TRY(c, hipFree(b.ptr));
    b.ptr = nullptr;
    b.size = 0;
    HIP_TRY(c, hipMalloc(&b.ptr, bytes + 256));
    b.size = bytes;
    return 0;
}

hipEvent_t get_event(vello_hip_ctx *c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int sync_all(vello_hip_ctx *c) {
    for (auto &l : c->lanes)
        if (l.stream) HIP_TRY(c, hipStreamSynchronize(l.stream));
    return 0;
}

// waits for the atlas uploads still in flight (before the atlas is freed / resized / the context goes away)
int sync_uploads(vello_hip_ctx *c) {
    if (c->upload_stream) HIP_TRY(c, hipStreamSynchronize(c->upload_stream));
    for (auto &st : c->staging) st.busy = false;
    return 0;
}

// the upload stream and its events (created on first use)
int ensure_upload_stream(vello_hip_ctx *c) {
    if (c->upload_stream) return 0;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->atlas_ready, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->lane_mark, hipEventDisableTiming));
    return 0;
}

// a pinned block of >= bytes that no DMA is reading
int acquire_staging(vello_hip_ctx *c, size_t bytes, Staging *&out) {
    size_t held = 0;
    for (auto &st : c->staging) {
        if (st.busy && hipEventQuery(st.done) == hipSuccess) st.busy = false;
        held += st.size;
    }
    for (auto &st : c->staging)
        if (!st.busy && st.size >= bytes) {
            out = &st;
            return 0;
        }
    if (held > ((size_t)256 << 20)) {  // bound the pinned memory: drain and start over
        int r = sync_uploads(c);
        if (r) return r;
        for (auto &st : c->staging) {
            (void)hipHostFree(st.host);
            (void)hipEventDestroy(st.done);
        }
        c->staging.clear();
    }
    Staging st;
    st.size = bytes < ((size_t)1 << 16) ? ((size_t)1 << 16) : bytes;
    HIP_TRY(c, hipHostMalloc(&st.host, st.size, hipHostMallocDefault));
    HIP_TRY(c, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    c->staging.push_back(st);
    out = &c->staging.back();
    return 0;
}

// fine's coverage scratch (sliced tiles): 64 words per FILL of a sliced tile.  A FILL is >= 5 command words, so 2 x the
// PTCL pool holds the fills of lists that add up to a sixth of the pool; tiles that do not fit are rendered unsliced
// (coarse only cuts what fits), so the scratch is an optimisation with a ceiling of its own -- 2^26 words, 256 MB per
// lane: it does not follow a PTCL pool grown into the gigabytes -- and a context without an MSAA mode, which never
// slices, holds none (it used to triple the PTCL footprint unconditionally).
constexpr uint32_t COV_CAP_MAX_WORDS = 1u << 26;
uint32_t cov_cap_words(const vello_hip_capacities &d, uint32_t aa_mask) {
    if ((aa_mask & (VELLO_HIP_AA_MASK_MSAA8 | VELLO_HIP_AA_MASK_MSAA16)) == 0u) return 0u;
    return d.ptcl >= COV_CAP_MAX_WORDS / 2u ? COV_CAP_MAX_WORDS : d.ptcl * 2u;
}

// words of coarse's tile bits: one per 8 tiles (three planes' bytes side by side, coarse.hip plane_window) + 2 of slack for
// the two-word windows read at the last tiles
uint32_t tile_bits_words(uint32_t tiles) { return (tiles + 63u) / 64u * 8u + 2u; }

// pool-capacity buffers of one lane (reference sizes: config.rs:398-408)
int alloc_lane_pools(vello_hip_ctx *c, Lane &l) {
    const vello_hip_capacities &d = c->caps;
    int r;
    if (!l.stream) HIP_TRY(c, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_LINES], (size_t)d.lines * sizeof(LineSoup)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_INFO_BIN_DATA], (size_t)d.bin_data * 4u))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_TILES], (size_t)d.tiles * sizeof(Tile)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_SEG_COUNTS], (size_t)d.seg_counts * sizeof(SegmentCount)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_SEGMENTS], (size_t)d.segments * sizeof(Segment)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_BLEND_SPILL], (size_t)d.blend_spill * 4u))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_PTCL], (size_t)d.ptcl * 4u))) return r;
    if ((r = ensure(c, l.tile_bits, (size_t)tile_bits_words(d.tiles) * 4u))) return r;
    if ((r = ensure(c, l.cov, (size_t)cov_cap_words(d, c->aa_mask) * 4u))) return r;
    return 0;
}

// bytes alloc_lane_pools asks for, per lane
size_t pool_bytes(const vello_hip_capacities &d, uint32_t aa_mask) {
    return (size_t)d.lines * sizeof(LineSoup) + (size_t)d.bin_data * 4u + (size_t)d.tiles * sizeof(Tile) +
           (size_t)d.seg_counts * sizeof(SegmentCount) + (size_t)d.segments * sizeof(Segment) + (size_t)d.blend_spill * 4u +
           (size_t)d.ptcl * 4u + (size_t)tile_bits_words(d.tiles) * 4u + (size_t)cov_cap_words(d, aa_mask) * 4u;
}

// Makes `d` the context's capacities, or leaves the context as it was: ensure() frees a buffer before it allocates the
// larger one, so after a failed hipMalloc the lane holds a null pool -- the old sizes are put back (they fitted before)
// and c->caps keeps describing what the lanes really hold.  Callers have drained the lanes.
int commit_caps(vello_hip_ctx *c, const vello_hip_capacities &d) {
    const vello_hip_capacities old = c->caps;
    c->caps = d;
    int r = 0;
    for (auto &l : c->lanes)
        if ((r = alloc_lane_pools(c, l))) break;
    if (!r) return 0;
    (void)hipGetLastError();
    const std::string err = c->last_error;
    c->caps = old;
    for (auto &l : c->lanes) {
        int r2 = alloc_lane_pools(c, l);
        if (r2) return r2;  // not even the old pools: nothing left to render into, the error stands
    }
    c->last_error = "growing the pools failed, capacities unchanged: " + err;
    return r;
}

SceneSlot &slot_of(vello_hip_ctx *c, Lane &l) { return l.use_own ? l.own : c->shared; }

// scene-dependent buffers of one lane
int alloc_lane_scene(vello_hip_ctx *c, Lane &l, const SceneSlot &sc) {
    const vello_hip_layout &L = sc.layout;
    int r;
    if ((r = ensure(c, l.zero_region, sc.zero_bytes))) return r;
    if (!l.front_sync.ptr) {
        if ((r = ensure(c, l.front_sync, 256))) return r;
        HIP_TRY(c, hipMemset(l.front_sync.ptr, 0, 256));
        HIP_TRY(c, hipStreamSynchronize(nullptr));  // (the lanes' streams do not wait for the null stream)
        l.front_sync_value = 0;
    }
    l.buf[VELLO_HIP_BUF_BUMP].ptr = l.zero_region.ptr;
    l.buf[VELLO_HIP_BUF_BUMP].size = sizeof(Control);  // (vello_hip_read_buffer: the bump allocators first, then the engine's own counters)
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_TAG_MONOIDS], (size_t)(sc.n_tag_words + 4u) * sizeof(TagMonoid)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_PATH_BBOXES], (size_t)(L.n_paths + 1u) * sizeof(PathBbox)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_DRAW_MONOIDS], (size_t)(L.n_draw_objects + 1u) * sizeof(DrawMonoid)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_CLIP_INP], (size_t)(L.n_clips + 1u) * sizeof(Clip)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_CLIP_BBOXES], (size_t)(L.n_clips + 1u) * sizeof(Bbox4)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_DRAW_BBOXES], (size_t)(L.n_paths + 1u) * sizeof(Bbox4)))) return r;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_PATHS], (size_t)(align_up(L.n_paths, 256u) + 256u) * sizeof(Path)))) return r;
    if ((r = ensure(c, l.clip_stack, clip_scratch_words(L.n_clips) * 4u))) return r;
    if ((r = ensure(c, l.coarse_el, (size_t)(L.n_draw_objects + 1u) * sizeof(CoarseEl)))) return r;
    if ((r = ensure(c, l.heavy_list, (size_t)(sc.n_tag_words + 1u) * 64u))) return r;  // 4 lists x 4 tags per word x u32
    {
        // one arc per stroked segment at most; a segment owns at least one word of path data (the tag stream is padded)
        const size_t n_tags = (size_t)sc.n_tag_words * 4u, n_data = (size_t)L.draw_tag_base - L.path_data_base;
        const uint32_t n_seg_max = (uint32_t)(n_tags < n_data ? n_tags : n_data);
        if ((r = ensure(c, l.arc_items, (size_t)(flatten_arc_shard_cap(n_seg_max, true) > flatten_arc_shard_cap(n_seg_max, false) ? flatten_arc_shard_cap(n_seg_max, true) : flatten_arc_shard_cap(n_seg_max, false)) * FLATTEN_ARC_SHARDS * 64u))) return r;
    }
    return 0;
}

// RenderConfig::new + BufferSizes::new, vello_encoding/src/config.rs:168-196, :363-435
int configure(vello_hip_ctx *c, const SceneSlot &sc, const vello_hip_render_params *p, Config &cfg) {
    if (!p || p->width == 0 || p->height == 0 || p->aa > VELLO_HIP_AA_MSAA16) {
        c->last_error = "invalid render params";
        return VELLO_HIP_E_INVALID;
    }
    if (((c->aa_mask >> p->aa) & 1u) == 0u) {
        // render.rs:566-568, :593-598: "shaders not configured to support AA mode"
        c->last_error = "AA mode was not enabled in vello_hip_create(aa_mask)";
        return VELLO_HIP_E_INVALID;
    }
    if (p->width > 0xffffu * TILE_WIDTH || p->height > 0xffffu * TILE_HEIGHT) {
        c->last_error = "target larger than 65535 tiles in one dimension";  // coarse packs tile coordinates in 16 bits
        return VELLO_HIP_E_INVALID;
    }
    std::memset(&cfg, 0, sizeof cfg);
    cfg.width_in_tiles = align_up(p->width, TILE_WIDTH) / TILE_WIDTH;
    cfg.height_in_tiles = align_up(p->height, TILE_HEIGHT) / TILE_HEIGHT;
    cfg.target_width = p->width;
    cfg.target_height = p->height;
    cfg.base_color = p->base_color;
    std::memcpy(&cfg.layout, &sc.layout, sizeof(Layout));
    uint32_t bin_data = c->caps.bin_data;
    if (bin_data <= sc.layout.bin_data_start) {
        c->last_error = "bin_data capacity smaller than the scene's info words";
        return VELLO_HIP_E_INVALID;
    }
    cfg.lines_size = c->caps.lines;
    cfg.binning_size = bin_data - sc.layout.bin_data_start;
    cfg.tiles_size = c->caps.tiles;
    cfg.seg_counts_size = c->caps.seg_counts;
    cfg.segments_size = c->caps.segments;
    cfg.blend_size = c->caps.blend_spill;
    cfg.ptcl_size = c->caps.ptcl;
    uint64_t initial_ptcl = (uint64_t)cfg.width_in_tiles * cfg.height_in_tiles * PTCL_INITIAL_ALLOC;
    if (initial_ptcl + PTCL_INCREMENT > cfg.ptcl_size) {
        c->last_error = "ptcl capacity smaller than the fixed per-tile allocation";
        return VELLO_HIP_E_INVALID;
    }
    return 0;
}

int prepare_frame(vello_hip_ctx *c, Lane &l, const vello_hip_render_params *p, void *out_device, size_t out_stride, Frame &f,
                  bool upload_cfg) {
    const SceneSlot &sc = slot_of(c, l);
    if (!sc.resident) {
        c->last_error = "no scene uploaded";
        return VELLO_HIP_E_INVALID;
    }
    int r;
    if (c->auto_grow && p && p->width && p->height) {
        // robust mode: a target with more tiles than the PTCL pool's fixed 64 words per tile (config.rs:408 sizes it for
        // ~1 Mpx) grows the pool instead of failing; the info words of the scene likewise must fit bin_data
        uint64_t tiles_xy = (uint64_t)(align_up(p->width, TILE_WIDTH) / TILE_WIDTH) * (align_up(p->height, TILE_HEIGHT) / TILE_HEIGHT);
        uint64_t need_ptcl = tiles_xy * PTCL_INITIAL_ALLOC + 64u * PTCL_INCREMENT;
        uint64_t need_bin = (uint64_t)sc.layout.bin_data_start + (1u << 16);
        if (need_ptcl > c->caps.ptcl || need_bin > c->caps.bin_data) {
            if (need_ptcl > 0xffff0000ull || need_bin > 0xffff0000ull) {
                c->last_error = "target / scene too large for 32-bit pool offsets";
                return VELLO_HIP_E_INVALID;
            }
            if ((r = sync_all(c))) return r;
            vello_hip_capacities d = c->caps;
            if (need_ptcl > d.ptcl) d.ptcl = (uint32_t)(need_ptcl + need_ptcl / 4u);
            if (need_bin > d.bin_data) d.bin_data = (uint32_t)(need_bin + need_bin / 4u);
            if ((r = commit_caps(c, d))) return r;
        }
    }
    r = configure(c, sc, p, f.cfg);
    if (r) return r;
    c->cfg = f.cfg;
    c->have_cfg = true;
    // size-dependent buffers
    uint32_t wb = (f.cfg.width_in_tiles + 15u) / 16u, hb = (f.cfg.height_in_tiles + 15u) / 16u;
    uint32_t aligned_n_bins = align_up(wb * hb, 256u);
    uint32_t binning_wgs = (sc.layout.n_draw_objects + 255u) / 256u;
    if ((r = ensure(c, l.buf[VELLO_HIP_BUF_BIN_HEADERS], (size_t)(binning_wgs * aligned_n_bins + 1u) * sizeof(BinHeader)))) return r;
    if (!out_device && (r = ensure(c, l.buf[VELLO_HIP_BUF_OUTPUT], (size_t)p->width * p->height * 4u))) return r;
    if ((r = ensure(c, l.tile_order, (size_t)f.cfg.width_in_tiles * f.cfg.height_in_tiles * FINE_WORK_BUCKETS * 4u))) return r;
    // fine launches one (at once exiting) workgroup per slice item it MIGHT be given in front of the tiles' workgroups:
    // as many as the scene asked for in an earlier frame (+ 1/8), a quarter of the tiles while that is unknown.  Coarse
    // cuts a tile into slices only if its items fit (the rest are rendered unsliced), so any number is correct.
    const uint32_t n_tiles_target = f.cfg.width_in_tiles * f.cfg.height_in_tiles;
    if ((r = ensure(c, l.slice_items, (size_t)n_tiles_target * sizeof(SliceItem)))) return r;
    if ((r = ensure(c, l.slice_counters, (size_t)n_tiles_target * 4u))) return r;
    {
        uint64_t want = sc.slice_demand < 0 ? n_tiles_target / 4u : (uint64_t)sc.slice_demand + (uint64_t)sc.slice_demand / 8u + 16u;
        if (want < 64u) want = 64u;
        f.slice_cap = want > n_tiles_target ? n_tiles_target : (uint32_t)want;
    }
    // kernels take the ConfigUniform by value (kernarg); the device copy only serves the test seam
    if (upload_cfg) HIP_TRY(c, hipMemcpy(c->config.ptr, &f.cfg, sizeof(Config), hipMemcpyHostToDevice));
    f.n_tag_words = sc.n_tag_words;
    f.n_scene_words = (uint32_t)(sc.scene_len / 4u);
    f.aa = p->aa;
    f.scene = (const uint32_t *)sc.scene.ptr;
    f.control = (Control *)l.zero_region.ptr;
    f.zero_bytes = (uint32_t)sc.zero_bytes;
    f.front_sync = (uint32_t *)l.front_sync.ptr;
    f.pathtag_state = (unsigned long long *)((char *)l.zero_region.ptr + sizeof(Control));
    f.draw_state = f.pathtag_state + (size_t)sc.n_pathtag_parts * 10u;
    f.tag_monoids = (TagMonoid *)l.buf[VELLO_HIP_BUF_TAG_MONOIDS].ptr;
    f.path_bboxes = (PathBbox *)l.buf[VELLO_HIP_BUF_PATH_BBOXES].ptr;
    f.lines = (LineSoup *)l.buf[VELLO_HIP_BUF_LINES].ptr;
    f.draw_monoids = (DrawMonoid *)l.buf[VELLO_HIP_BUF_DRAW_MONOIDS].ptr;
    f.info_bin_data = (uint32_t *)l.buf[VELLO_HIP_BUF_INFO_BIN_DATA].ptr;
    f.clip_inp = (Clip *)l.buf[VELLO_HIP_BUF_CLIP_INP].ptr;
    f.clip_bboxes = (Bbox4 *)l.buf[VELLO_HIP_BUF_CLIP_BBOXES].ptr;
    f.draw_bboxes = (Bbox4 *)l.buf[VELLO_HIP_BUF_DRAW_BBOXES].ptr;
    f.bin_headers = (BinHeader *)l.buf[VELLO_HIP_BUF_BIN_HEADERS].ptr;
    f.paths = (Path *)l.buf[VELLO_HIP_BUF_PATHS].ptr;
    f.tiles = (Tile *)l.buf[VELLO_HIP_BUF_TILES].ptr;
    f.seg_counts = (SegmentCount *)l.buf[VELLO_HIP_BUF_SEG_COUNTS].ptr;
    f.segments = (Segment *)l.buf[VELLO_HIP_BUF_SEGMENTS].ptr;
    f.ptcl = (uint32_t *)l.buf[VELLO_HIP_BUF_PTCL].ptr;
    f.blend_spill = (uint32_t *)l.buf[VELLO_HIP_BUF_BLEND_SPILL].ptr;
    f.clip_stack = (uint32_t *)l.clip_stack.ptr;
    f.coarse_el = (CoarseEl *)l.coarse_el.ptr;
    f.tile_bits = (uint32_t *)l.tile_bits.ptr;
    f.tile_order = (uint32_t *)l.tile_order.ptr;
    f.slice_items = (SliceItem *)l.slice_items.ptr;
    f.slice_counters = (uint32_t *)l.slice_counters.ptr;
    f.cov = (uint32_t *)l.cov.ptr;
    f.cov_cap = cov_cap_words(c->caps, c->aa_mask);
    f.slice_fills = f.slice_min_fills = 0u;
    l.slices_on = p->aa != 0u;
    if (p->aa != 0u) {
        const bool forced = (c->debug_flags & VELLO_HIP_DEBUG_FINE_SLICES) != 0u;
        const bool alone = c->n_active == 1u;
        f.slice_fills = forced ? FINE_SLICE_FILLS_FORCED : (alone ? FINE_SLICE_FILLS : FINE_SLICE_FILLS_IN_FLIGHT);
        f.slice_min_fills = forced ? FINE_SLICE_MIN_FILLS_FORCED : (alone ? FINE_SLICE_MIN_FILLS : FINE_SLICE_MIN_FILLS_IN_FLIGHT);
    }
    f.heavy_list = (uint32_t *)l.heavy_list.ptr;
    f.arc_items = (uint32_t *)l.arc_items.ptr;
    if (out_device) {
        f.output = (uint8_t *)out_device;
        f.out_stride = out_stride ? out_stride : (size_t)p->width * 4u;
    } else {
        f.output = (uint8_t *)l.buf[VELLO_HIP_BUF_OUTPUT].ptr;
        f.out_stride = (size_t)p->width * 4u;
    }
    f.ramps = sc.n_ramps ? (const uint32_t *)sc.ramps.ptr : nullptr;
    f.n_ramps = sc.n_ramps;
    f.brushes = sc.brushes || c->force_brushes;
    f.no_cull = (c->debug_flags & VELLO_HIP_DEBUG_NO_CULL) != 0u;
    f.sequential_clip = (c->debug_flags & VELLO_HIP_DEBUG_SEQ_CLIP) != 0u;
    f.stroke_kernel_min_lines = (c->debug_flags & VELLO_HIP_DEBUG_STROKE_KERNEL) != 0u ? 0u : FLATTEN_STROKE_KERNEL_MIN_LINES;
    f.path_count_small = sc.soup_lines >= 0 && sc.soup_lines < PATH_COUNT_SMALL_MAX_LINES;
    f.flatten_side_by_side = c->n_active == 1u;
    f.launch_stroke_kernel = sc.stroke_lines < 0 || (uint64_t)sc.stroke_lines >= f.stroke_kernel_min_lines;
    // Which kernels take flatten's heavy list (flatten_walk.inc).  The wave-cooperative walk pays where a lane alone would loop
    // long: a list short enough for a wave per entry (<= 4 096: the tiger's 2 300 curves, a circle's four), or one that is mostly
    // stroked curves (mmark: two offset curves a walk, a dozen lines a turn); the road map's blobs, joins and caps are 10 % faster
    // with every lane on its own and the fp64 routines inline.  Known from a finished frame of the scene; before there is one, by
    // the scene's size (a segment owns at least one word of path data).
    {
        const uint64_t n_data = sc.layout.draw_tag_base - sc.layout.path_data_base, n_tags4 = (uint64_t)sc.n_tag_words * 4u;
        const uint64_t n_seg_max = n_tags4 < n_data ? n_tags4 : n_data;
        bool coop = n_seg_max <= 16384u;
        if (sc.heavy_curves >= 0 && sc.stroke_lines >= 0) {
            const uint64_t nc = (uint64_t)sc.heavy_curves, ns = (uint64_t)sc.heavy_strokes;
            const uint64_t nl = (uint64_t)sc.stroke_lines < f.stroke_kernel_min_lines ? (uint64_t)sc.stroke_lines : 0u;
            const uint64_t n_heavy = nc + ns + nl;
            coop = n_heavy <= 4096u || ns * 2u > n_heavy + nc * 2u;
        }
        if (c->debug_flags & VELLO_HIP_DEBUG_FLATTEN_COOP) coop = true;
        if (c->debug_flags & VELLO_HIP_DEBUG_FLATTEN_ALONE) coop = false;
        f.flatten_coop = coop;
    }
    l.frame_generation = sc.generation;
    f.atlas = c->atlas_w ? (const uint32_t *)c->atlas.ptr : nullptr;
    f.atlas_w = c->atlas_w;
    f.atlas_h = c->atlas_h;
    f.mask_lut8 = (const uint32_t *)c->mask8.ptr;
    f.mask_lut16 = (const uint32_t *)c->mask16.ptr;
    if (l.atlas_epoch_seen != c->atlas_epoch) {  // atlas uploads enqueued since this lane's last frame come first
        HIP_TRY(c, hipStreamWaitEvent(l.stream, c->atlas_ready, 0));
        l.atlas_epoch_seen = c->atlas_epoch;
    }
    l.used = true;
    return 0;
}

// k_front's grid-barrier counter (Lane::front_sync) only ever grows, and the host keeps its value: a launch's barriers wait for
// sync_base + k * (its workgroups).  The two must never drift apart: if a launch is rejected the device never adds, if a
// barrier's spin bound trips the workgroups stop waiting -- either way the counter goes back to zero on the lane's stream, behind
// whatever was enqueued, and the host's copy with it.
static void reset_front_sync(Lane &l, hipStream_t st) {
    if (l.front_sync.ptr) (void)hipMemsetAsync(l.front_sync.ptr, 0, 256, st);
    l.front_sync_value = 0u;
}
// After a k_front launch: the host's copy of the counter advances only once the launch was accepted.
static int front_accepted(vello_hip_ctx *c, Lane &l, hipStream_t st, uint32_t add) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        reset_front_sync(l, st);
        c->last_error = std::string("k_front launch: ") + hipGetErrorString(e);
        return VELLO_HIP_E_HIP;
    }
    l.front_sync_value += add;
    c->fused_launches++;
    return 0;
}

int run_stage_range(vello_hip_ctx *c, Lane &l, const Frame &f_in, int first, int last) {
    hipStream_t st = l.stream;
    // coarse decides which tiles are cut into slices from the number of slice blocks fine is going to launch: a FINE that
    // runs in a later call than its COARSE (vello_hip_run_stages) launches with the number COARSE was given, whatever
    // the scene's demand is known to be by then
    Frame f = f_in;
    if (first > VELLO_HIP_STAGE_COARSE) {
        if (last >= VELLO_HIP_STAGE_FINE && l.slice_fills_coarse != f.slice_fills) {
            // (coarse's sliced tiles are not in the work buckets: an area-AA fine would never render them)
            c->last_error = "vello_hip_run_stages: fine's AA mode must be of the same kind (area / MSAA) as the one coarse ran with";
            return VELLO_HIP_E_INVALID;
        }
        f.slice_cap = l.slice_cap_coarse;
    } else if (last >= VELLO_HIP_STAGE_COARSE) {
        l.slice_cap_coarse = f.slice_cap;
        l.slice_fills_coarse = f.slice_fills;
    }
    // Small scenes: consecutive stages as one launch (k_front, flatten.hip) -- A: [zero fill | pathtag scan | flatten's light pass
    // + draw scan], B: [binning | tile_alloc]; a scene of a few dozen segments: everything up to tile_alloc.  Not for a stage that
    // is being timed on its own.
    bool fuse_a = false, fuse_b = false, fuse_all = false;
    // Any scene: binning and tile_alloc as one launch (binning.hip k_binning_tile_alloc) when the range holds both and neither is
    // timed on its own
    const bool pair_b = (c->debug_flags & VELLO_HIP_DEBUG_NO_FUSION) == 0u && first <= VELLO_HIP_STAGE_BINNING && last >= VELLO_HIP_STAGE_TILE_ALLOC &&
                        (c->prof_mask & ((1u << VELLO_HIP_STAGE_BINNING) | (1u << VELLO_HIP_STAGE_TILE_ALLOC))) == 0u;
    if ((c->debug_flags & VELLO_HIP_DEBUG_NO_FUSION) == 0u && f.n_tag_words * 4u <= FRONT_MAX_TAGS &&
        f.cfg.layout.n_draw_objects <= FRONT_MAX_DRAW_OBJECTS && f.cfg.layout.n_paths <= FRONT_MAX_DRAW_OBJECTS) {
        const uint32_t pm = c->prof_mask;
        const uint32_t mask_a = (1u << VELLO_HIP_STAGE_PATHTAG_SCAN) | (1u << VELLO_HIP_STAGE_FLATTEN) | (1u << VELLO_HIP_STAGE_DRAW_SCAN);
        const uint32_t mask_b = (1u << VELLO_HIP_STAGE_BINNING) | (1u << VELLO_HIP_STAGE_TILE_ALLOC);
        fuse_a = first == VELLO_HIP_STAGE_PATHTAG_SCAN && last >= VELLO_HIP_STAGE_FLATTEN && (pm & mask_a) == 0u;
        fuse_b = first <= VELLO_HIP_STAGE_BINNING && last >= VELLO_HIP_STAGE_TILE_ALLOC && (pm & mask_b) == 0u;
        // (the heavy list aboard: the kernels of the cooperative walk, no stroke workgroups -- not when a debug flag asks for others)
        fuse_all = fuse_a && fuse_b && (pm & (1u << VELLO_HIP_STAGE_CLIP)) == 0u && f.cfg.layout.n_clips == 0u &&
                   flatten_n_seg_max(f) <= FRONT_TINY_SEGMENTS && f.flatten_coop && f.stroke_kernel_min_lines != 0u;
    }
    for (int s = first; s <= last; s++) {
        bool prof = ((c->prof_mask >> s) & 1u) != 0u;
        uint32_t front_add = 0u;  // what a k_front launch of this stage adds to the lane's barrier counter (0: one workgroup)
        bool front_launched = false;
        Lane::EvPair ev{s, nullptr, nullptr, {nullptr, nullptr}};
        if (prof) {
            ev.a = get_event(c);
            ev.b = get_event(c);
            if (s == VELLO_HIP_STAGE_FLATTEN || s == VELLO_HIP_STAGE_COARSE) ev.mid[0] = get_event(c);
            if (s == VELLO_HIP_STAGE_FLATTEN) ev.mid[1] = get_event(c);
            HIP_TRY(c, hipEventRecord(ev.a, st));
        }
        // experiment: a stage named in the debug flags runs for ONE frame at a time -- its launches wait for the same stage of the
        // frame enqueued before (on another lane) to be done
        const bool exclusive = c->n_active > 1u && ((c->debug_flags >> (VELLO_HIP_DEBUG_EXCLUSIVE_SHIFT + s)) & 1u) != 0u;
        if (exclusive) {
            if (!c->stage_turn[s]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_turn[s], hipEventDisableTiming));
            else HIP_TRY(c, hipStreamWaitEvent(st, c->stage_turn[s], 0));
        }
        switch (s) {
        case VELLO_HIP_STAGE_PATHTAG_SCAN:
            // render.rs:313 clears `bump`; the same memset resets both look-back states and tickets
            l.flatten_ran = false;
            if (fuse_all) {
                front_add = launch_front(f, st, FRONT_ZERO | FRONT_PATHTAG | FRONT_LIGHT | FRONT_HEAVY | FRONT_BINNING | FRONT_TILE_ALLOC, true, l.front_sync_value);
                front_launched = true;
                break;
            }
            if (fuse_a) break;  // (with FLATTEN's first launch)
            HIP_TRY(c, hipMemsetAsync(l.zero_region.ptr, 0, slot_of(c, l).zero_bytes, st));
            launch_pathtag_scan(f, st);
            break;
        case VELLO_HIP_STAGE_FLATTEN:
            l.flatten_ran = true;
            if (fuse_all) break;
            // (a range that goes on to DRAW_SCAN: that stage's workgroups ride in flatten's first launch)
            if (fuse_a) {
                // (k_front's barrier counter advances on the host only once the launch was accepted: ADVICE r5)
                const uint32_t add = launch_front(f, st, FRONT_ZERO | FRONT_PATHTAG | FRONT_LIGHT, last >= VELLO_HIP_STAGE_DRAW_SCAN, l.front_sync_value);
                if (int r = front_accepted(c, l, st, add)) return r;
            }
            launch_flatten(f, st, prof ? ev.mid : nullptr, last >= VELLO_HIP_STAGE_DRAW_SCAN, fuse_a);
            break;
        case VELLO_HIP_STAGE_DRAW_SCAN:
            if (first > VELLO_HIP_STAGE_FLATTEN) launch_draw_scan(f, st);  // (else: done beside k_flatten_light)
            break;
        case VELLO_HIP_STAGE_CLIP: launch_clip(f, st); break;
        case VELLO_HIP_STAGE_BINNING:
            if (fuse_all) break;
            if (fuse_b) {
                if (f.cfg.layout.n_draw_objects != 0u || f.cfg.layout.n_paths != 0u) {
                    front_add = launch_front(f, st, FRONT_BINNING | FRONT_TILE_ALLOC, false, l.front_sync_value);
                    front_launched = true;
                }
            } else if (pair_b) {
                launch_binning_tile_alloc(f, st);
            } else {
                launch_binning(f, st);
            }
            break;
        case VELLO_HIP_STAGE_TILE_ALLOC:
            if (!fuse_all && !fuse_b && !pair_b) launch_tile_alloc(f, st);
            break;
        case VELLO_HIP_STAGE_PATH_COUNT: launch_path_count(f, st); break;
        case VELLO_HIP_STAGE_BACKDROP: launch_backdrop(f, st); break;
        case VELLO_HIP_STAGE_COARSE: launch_coarse(f, st, prof ? ev.mid : nullptr); break;
        case VELLO_HIP_STAGE_PATH_TILING: launch_path_tiling(f, st); break;
        case VELLO_HIP_STAGE_FINE: launch_fine(f, st); break;
        default: return VELLO_HIP_E_INVALID;
        }
        if (front_launched) {
            if (int r = front_accepted(c, l, st, front_add)) return r;
        } else {
            HIP_TRY(c, hipGetLastError());
        }
        if (exclusive) HIP_TRY(c, hipEventRecord(c->stage_turn[s], st));
        if (prof) {
            HIP_TRY(c, hipEventRecord(ev.b, st));
            l.events.push_back(ev);
        }
    }
    return 0;
}

int drain_events(vello_hip_ctx *c) {
    for (auto &l : c->lanes) {
        for (auto &ev : l.events) {
            float ms = 0.f;
            HIP_TRY(c, hipEventSynchronize(ev.b));
            HIP_TRY(c, hipEventElapsedTime(&ms, ev.a, ev.b));
            c->stage_ms[ev.stage] += ms;
            c->stage_count[ev.stage] += 1;
            if (ev.mid[0]) {  // a -> mid[0] (-> mid[1]) -> b: the stage's kernels one by one
                hipEvent_t pts[4] = {ev.a, ev.mid[0], ev.mid[1] ? ev.mid[1] : ev.b, ev.b};
                const int n_k = ev.mid[1] ? 3 : 2;
                for (int k = 0; k < n_k; k++) {
                    float kms = 0.f;
                    if (hipEventElapsedTime(&kms, pts[k], pts[k + 1]) == hipSuccess) c->kernel_ms[ev.stage][k] += kms;
                }
                c->kernel_count[ev.stage] += 1;
                c->event_pool.push_back(ev.mid[0]);
                if (ev.mid[1]) c->event_pool.push_back(ev.mid[1]);
            }
            c->event_pool.push_back(ev.a);
            c->event_pool.push_back(ev.b);
        }
        l.events.clear();
    }
    return 0;
}

DevBuf *find_buf(vello_hip_ctx *c, int id) {
    if (id == VELLO_HIP_BUF_SCENE) return &slot_of(c, c->lanes[c->last_lane]).scene;
    if (id == VELLO_HIP_BUF_CONFIG) return &c->config;
    return &c->lanes[c->last_lane].buf[id];
}

// vello_encoding/src/mask.rs:11-98
const uint8_t PATTERN8[8] = {0, 5, 3, 7, 1, 4, 6, 2};
const uint8_t PATTERN16[16] = {1, 8, 4, 11, 15, 7, 3, 12, 0, 9, 5, 13, 2, 10, 6, 14};
uint32_t one_mask_n(double slope, double translation, bool is_pos, const uint8_t *pat, int n) {
    if (is_pos) translation = 1. - translation;
    uint32_t result = 0;
    double inv = 1.0 / (double)n;
    for (int i = 0; i < n; i++) {
        double y = ((double)i + 0.5) * inv;
        double x = ((double)pat[i] + 0.5) * inv;
        if (!is_pos) y = 1. - y;
        if ((x - (1.0 - translation)) * (1. - slope) - (y - translation) * slope >= 0.) result |= 1u << i;
    }
    return result;
}

}  // namespace

extern "C" {

void vello_hip_make_mask_lut(uint8_t out[1024]) {
    const int W = 32, H = 32, HALF = 16;
    for (int i = 0; i < W * H; i++) {
        int u = i % W, v = i / W;
        double y = ((double)(v % HALF) + 0.5) * (1.0 / (double)HALF);
        double x = ((double)u + 0.5) * (1.0 / (double)W);
        out[i] = (uint8_t)one_mask_n(y, x, v >= HALF, PATTERN8, 8);
    }
}

void vello_hip_make_mask_lut_16(uint8_t out[8192]) {
    const int W = 64, H = 64, HALF = 32;
    for (int i = 0; i < W * H; i++) {
        int u = i % W, v = i / W;
        double y = ((double)(v % HALF) + 0.5) * (1.0 / (double)HALF);
        double x = ((double)u + 0.5) * (1.0 / (double)W);
        uint32_t m = one_mask_n(y, x, v >= HALF, PATTERN16, 16);
        out[2 * i] = (uint8_t)(m & 0xff);
        out[2 * i + 1] = (uint8_t)(m >> 8);
    }
}

const char *vello_hip_stage_name(int stage) {
    if (stage < 0 || stage >= VELLO_HIP_STAGE_COUNT) return "?";
    return kStageNames[stage];
}

const char *vello_hip_last_error(vello_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int vello_hip_create(int device, uint32_t aa_mask, const vello_hip_capacities *caps, vello_hip_ctx **out) {
    if (!out) return VELLO_HIP_E_INVALID;
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
        g_create_error = std::string("no usable HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                         "); the engine has no CPU fallback";
        return VELLO_HIP_E_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_create_error = "hipSetDevice failed";
        return VELLO_HIP_E_NO_DEVICE;
    }
    // The process's FIRST hardware queue is a poor place for a lane: with four frames in flight a context whose lane 0 sits on it
    // renders 8 % fewer frames per second than one whose lanes come second to fifth (profiles/r06_queue_order.txt: 2 110 against
    // 2 300 on the road map, rocprofv3's Queue_Id per stream).  HIP hands out hardware queues in the order streams are created,
    // and creates the null stream's lazily -- so the null stream is made to take its queue here, before the lanes' streams exist
    // (a caller that has used the device before gets the same order by itself).
    {
        void *scratch = nullptr;
        if (hipMalloc(&scratch, 256) == hipSuccess) {
            (void)hipMemsetAsync(scratch, 0, 256, nullptr);
            (void)hipStreamSynchronize(nullptr);
            (void)hipFree(scratch);
        }
        (void)hipGetLastError();
    }
    vello_hip_ctx *c = new vello_hip_ctx();
    c->device = device;
    c->aa_mask = aa_mask ? (aa_mask & VELLO_HIP_AA_MASK_ALL) : VELLO_HIP_AA_MASK_ALL;
    // reference pool sizes, vello_encoding/src/config.rs:398-408
    vello_hip_capacities d{1u << 21, 1u << 18, 1u << 21, 1u << 21, 1u << 21, 1u << 20, 1u << 23};
    if (caps) {
        if (caps->lines) d.lines = caps->lines;
        if (caps->bin_data) d.bin_data = caps->bin_data;
        if (caps->tiles) d.tiles = caps->tiles;
        if (caps->seg_counts) d.seg_counts = caps->seg_counts;
        if (caps->segments) d.segments = caps->segments;
        if (caps->blend_spill) d.blend_spill = caps->blend_spill;
        if (caps->ptcl) d.ptcl = caps->ptcl;
    }
    // path_tiling writes segments[] unguarded by a failure bit only when both pools match (coarse.wgsl:166)
    if (d.segments < d.seg_counts) d.segments = d.seg_counts;
    c->caps = d;
    auto fail = [&](const char *what) {
        g_create_error = std::string(what) + ": " + c->last_error;
        vello_hip_destroy(c);
        return VELLO_HIP_E_HIP;
    };
    if (int le = enable_coarse_lds()) {
        c->last_error = std::string("hipFuncSetAttribute(k_coarse, MaxDynamicSharedMemorySize): ") + hipGetErrorString((hipError_t)le);
        return fail("coarse LDS opt-in");
    }
    c->lanes.resize(1);
    if (alloc_lane_pools(c, c->lanes[0])) return fail("pool allocation");
    if (ensure(c, c->config, sizeof(Config))) return fail("config");
    if (ensure(c, c->mask8, 1024) || ensure(c, c->mask16, 8192)) return fail("mask lut");
    {
        std::vector<uint8_t> l8(1024), l16(8192);
        vello_hip_make_mask_lut(l8.data());
        vello_hip_make_mask_lut_16(l16.data());
        if (hipMemcpy(c->mask8.ptr, l8.data(), 1024, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->mask16.ptr, l16.data(), 8192, hipMemcpyHostToDevice) != hipSuccess)
            return fail("mask lut upload");
    }
    // Pre-warm: the first launch of a kernel loads its code object and, for the kernels with a scratch segment (fine, the
    // heavy flattener), makes the runtime allocate scratch for the queue -- ~20 ms that would otherwise land in the
    // caller's first frame.  One frame of an empty scene per enabled AA mode and fine specialisation.
    {
        std::vector<uint8_t> empty(1024u * 4u, 0);  // 1024 zero tag words = 4096 padded path tags, nothing else
        vello_hip_layout lay{};
        lay.path_tag_base = 0u;
        lay.path_data_base = lay.draw_tag_base = lay.draw_data_base = lay.transform_base = lay.style_base = 1024u;
        if (vello_hip_upload_scene(c, empty.data(), empty.size(), &lay, nullptr, 0) == VELLO_HIP_OK) {
            for (uint32_t aa = 0; aa < 3u; aa++) {
                if (((c->aa_mask >> aa) & 1u) == 0u) continue;
                for (int brushes = 0; brushes < 2; brushes++) {
                    c->force_brushes = brushes != 0;
                    vello_hip_render_params rp{16u, 16u, 0u, aa};
                    (void)vello_hip_render_resident(c, &rp, nullptr, 0);
                }
            }
            c->force_brushes = false;
            (void)sync_all(c);
            c->lanes[0].used = false;
            c->have_cfg = false;
        }
        c->shared.resident = false;  // the caller uploads its own scene
        c->last_error.clear();
    }
    *out = c;
    return VELLO_HIP_OK;
}

void vello_hip_destroy(vello_hip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto &l : c->lanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        for (auto &ev : l.events) {
            (void)hipEventDestroy(ev.a);
            (void)hipEventDestroy(ev.b);
        }
        for (int i = 0; i < VELLO_HIP_BUF_COUNT; i++)
            if (l.buf[i].ptr && i != VELLO_HIP_BUF_BUMP) (void)hipFree(l.buf[i].ptr);
        if (l.zero_region.ptr) (void)hipFree(l.zero_region.ptr);
        if (l.front_sync.ptr) (void)hipFree(l.front_sync.ptr);
        if (l.clip_stack.ptr) (void)hipFree(l.clip_stack.ptr);
        if (l.coarse_el.ptr) (void)hipFree(l.coarse_el.ptr);
        if (l.tile_bits.ptr) (void)hipFree(l.tile_bits.ptr);
        if (l.tile_order.ptr) (void)hipFree(l.tile_order.ptr);
        for (DevBuf *b : {&l.slice_items, &l.slice_counters, &l.cov})
            if (b->ptr) (void)hipFree(b->ptr);
        if (l.heavy_list.ptr) (void)hipFree(l.heavy_list.ptr);
        if (l.arc_items.ptr) (void)hipFree(l.arc_items.ptr);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->upload_stream) {
        (void)hipStreamSynchronize(c->upload_stream);
        for (auto &st : c->staging) {
            (void)hipHostFree(st.host);
            (void)hipEventDestroy(st.done);
        }
        (void)hipEventDestroy(c->atlas_ready);
        (void)hipEventDestroy(c->lane_mark);
        for (auto &e : c->stage_turn)
            if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(c->upload_stream);
    }
    if (c->frame_done) (void)hipEventDestroy(c->frame_done);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto &l : c->lanes)
        for (DevBuf *b : {&l.own.scene, &l.own.ramps})
            if (b->ptr) (void)hipFree(b->ptr);
    for (DevBuf *b : {&c->shared.scene, &c->shared.ramps, &c->config, &c->mask8, &c->mask16, &c->atlas})
        if (b->ptr) (void)hipFree(b->ptr);
    delete c;
}

int vello_hip_set_frames_in_flight(vello_hip_ctx *c, uint32_t n) {
    if (!c || n < 1 || n > 8) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = sync_all(c);
    if (r) return r;
    size_t old = c->lanes.size();
    if (n > old) {
        c->lanes.resize(n);
        for (size_t i = old; i < n; i++) {
            if ((r = alloc_lane_pools(c, c->lanes[i]))) return r;
            if (c->shared.resident && (r = alloc_lane_scene(c, c->lanes[i], c->shared))) return r;
        }
    }
    if ((c->n_active == 1u) != (n == 1u)) {
        // fine's slicing threshold differs between one frame at a time and frames in flight (engine.h FINE_SLICE_MIN_FILLS*): the
        // slice items a frame of the scene was seen to ask for say nothing about the other mode -- unknown again (the default
        // capacity, and the next finished frame is read back)
        c->shared.slice_demand = -1;
        for (auto &l : c->lanes) l.own.slice_demand = -1;
    }
    c->n_active = n;  // shrinking keeps the extra lanes' buffers; only the rotation changes
    c->next_lane = 0;
    c->last_lane = 0;
    return VELLO_HIP_OK;
}

// Validates the layout, sizes the slot and copies scene + ramps on `st`; returns once the source buffers may be
// reused (they are caller-owned only for the duration of the call, recording.rs:124-129).
static int load_slot(vello_hip_ctx *c, SceneSlot &sc, hipStream_t st, const uint8_t *scene, size_t scene_len,
                     const vello_hip_layout *layout, const uint32_t *ramps, uint32_t n_ramps) {
    if ((!scene && scene_len) || !layout || (scene_len & 3u)) return VELLO_HIP_E_INVALID;
    const vello_hip_layout &L = *layout;
    size_t words = scene_len / 4u;
    if (L.path_tag_base > L.path_data_base || L.path_data_base > L.draw_tag_base || L.draw_tag_base > L.draw_data_base ||
        L.draw_data_base > L.transform_base || L.transform_base > L.style_base || L.style_base > words ||
        (size_t)L.draw_tag_base + L.n_draw_objects > words) {
        c->last_error = "layout does not describe the scene buffer";
        return VELLO_HIP_E_INVALID;
    }
    if ((((size_t)L.path_data_base - L.path_tag_base) * 4u) % 1024u != 0u) {
        c->last_error = "path tag stream is not padded to 4*256 tags (resolve.rs:622-639)";
        return VELLO_HIP_E_INVALID;
    }
    int r;
    // 64 B of slack: flatten reads tag ix+1 and the (wrapped) style word of pre-style tags speculatively
    if ((r = ensure(c, sc.scene, scene_len + 64))) return r;
    sc.layout = L;
    sc.scene_len = scene_len;
    uint32_t n_path_tags = (L.path_data_base - L.path_tag_base) * 4u;
    sc.n_tag_words = align_up(n_path_tags, 1024u) / 4u;
    sc.n_pathtag_parts = (sc.n_tag_words + PATHTAG_PART_WORDS - 1u) / PATHTAG_PART_WORDS;
    if (sc.n_pathtag_parts == 0) sc.n_pathtag_parts = 1;
    sc.n_draw_parts = (L.n_draw_objects + DRAW_PART - 1u) / DRAW_PART;
    sc.zero_bytes = sizeof(Control) + ((size_t)sc.n_pathtag_parts * 10u + (size_t)sc.n_draw_parts * 8u) * 8u;
    // Which fine specialisation the scene needs: any draw tag other than COLOR / BEGIN_CLIP / END_CLIP / NOP
    // (draw.rs:15-51) makes coarse emit a gradient, image or blur command.
    sc.brushes = false;
    sc.stroke_lines = -1;
    sc.heavy_curves = sc.heavy_strokes = -1;
    sc.soup_lines = -1;
    sc.slice_demand = -1;
    sc.generation += 1u;
    {
        // The same pass checks what draw_leaf / clip_leaf will index with (shared/drawtag.wgsl:47-54: bit 0 = clip,
        // bits 2-4 = draw data words, bits 6-9 = info words).  WebGPU's robust buffer access absorbs an inconsistent
        // stream upstream; here it must be refused.  (The path tag stream is checked by the pathtag scan, on the GPU.)
        const uint32_t *words_p = reinterpret_cast<const uint32_t *>(scene);
        uint64_t draw_data_words = 0, info_words = 0, clip_tags = 0;
        for (uint32_t i = 0; i < L.n_draw_objects; i++) {
            uint32_t t = words_p[L.draw_tag_base + i];
            if (t != DRAWTAG_FILL_COLOR && t != DRAWTAG_BEGIN_CLIP && t != DRAWTAG_END_CLIP && t != DRAWTAG_NOP) sc.brushes = true;
            clip_tags += t & 1u;
            draw_data_words += (t >> 2) & 0x7u;
            info_words += (t >> 6) & 0xfu;
        }
        // clip_tags == n_clips, not <=: k_clip walks n_clips entries of clip_inp and draw_leaf writes one per clip tag
        // (resolve counts exactly the BEGIN/END_CLIP tags below n_draw_objects: the END_CLIPs it appends for unclosed
        // layers lie behind them, resolve.rs:139-141); fewer tags would leave entries uninitialised
        if (draw_data_words > (uint64_t)(L.transform_base - L.draw_data_base) || info_words > L.bin_data_start || clip_tags != L.n_clips ||
            L.n_draw_objects > L.n_paths) {
            c->last_error = "draw tags need more draw data / info words / paths than the layout provides, or their clip count differs from n_clips";
            return VELLO_HIP_E_INVALID;
        }
    }
    if (scene_len) HIP_TRY(c, hipMemcpyAsync(sc.scene.ptr, scene, scene_len, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync((char *)sc.scene.ptr + scene_len, 0, 64, st));
    sc.n_ramps = 0;
    if (ramps && n_ramps) {
        if ((r = ensure(c, sc.ramps, (size_t)n_ramps * 512u * 4u))) return r;
        HIP_TRY(c, hipMemcpyAsync(sc.ramps.ptr, ramps, (size_t)n_ramps * 512u * 4u, hipMemcpyHostToDevice, st));
        sc.n_ramps = n_ramps;
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    sc.resident = true;
    return VELLO_HIP_OK;
}

int vello_hip_upload_scene(vello_hip_ctx *c, const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                           const uint32_t *ramps, uint32_t n_ramps) {
    if (!c) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int r;
    // frames still in flight read the old scene
    if ((r = sync_all(c))) return r;
    if ((r = load_slot(c, c->shared, c->lanes[0].stream, scene, scene_len, layout, ramps, n_ramps))) return r;
    for (auto &l : c->lanes) {
        l.use_own = false;
        if ((r = alloc_lane_scene(c, l, c->shared))) return r;
    }
    return VELLO_HIP_OK;
}

// The pipelined form of vello_hip_render for animations: every frame brings its own scene.  The scene goes into
// the private slot of the next lane of the rotation (only THAT lane's previous frame is waited for), the frame is
// enqueued, and the call returns; other lanes keep rendering while the next scene crosses PCIe.
int vello_hip_render_frame(vello_hip_ctx *c, const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                           const vello_hip_render_params *params, const uint32_t *ramps, uint32_t n_ramps, void *out_device,
                           size_t out_stride) {
    if (!c || !params) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t li = c->next_lane % c->n_active;
    Lane &l = c->lanes[li];
    HIP_TRY(c, hipStreamSynchronize(l.stream));
    int r = load_slot(c, l.own, l.stream, scene, scene_len, layout, ramps, n_ramps);
    if (r) return r;
    l.use_own = true;
    if ((r = alloc_lane_scene(c, l, l.own))) return r;
    c->next_lane = (li + 1u) % c->n_active;
    c->last_lane = li;
    Frame f;
    if ((r = prepare_frame(c, l, params, out_device, out_stride, f, false))) return r;
    return run_stage_range(c, l, f, 0, VELLO_HIP_STAGE_FINE);
}

int vello_hip_resize_image_atlas(vello_hip_ctx *c, uint32_t width, uint32_t height) {
    if (!c || width > 0xffffu || height > 0xffffu) return VELLO_HIP_E_INVALID;  // DrawImage packs xy / extents in 16 bits
    HIP_TRY(c, hipSetDevice(c->device));
    int r = sync_all(c);
    if (r) return r;
    if ((r = sync_uploads(c))) return r;
    c->atlas_w = c->atlas_h = 0;
    if (width == 0 || height == 0) return VELLO_HIP_OK;
    size_t bytes = (size_t)width * height * 4u;
    if ((r = ensure(c, c->atlas, bytes))) return r;
    // Every writer of the atlas is ordered on the upload stream: hipMemset on the null stream is asynchronous to the host
    // for device memory and the (non-blocking) upload stream does not synchronise with it, so a clear issued there could
    // land AFTER the uploads that follow this call.  Frames wait for `atlas_ready` like they do after an upload.
    if ((r = ensure_upload_stream(c))) return r;
    HIP_TRY(c, hipMemsetAsync(c->atlas.ptr, 0, bytes, c->upload_stream));
    HIP_TRY(c, hipEventRecord(c->atlas_ready, c->upload_stream));
    c->atlas_epoch += 1u;
    c->atlas_w = width;
    c->atlas_h = height;
    return VELLO_HIP_OK;
}

int vello_hip_write_image(vello_hip_ctx *c, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const uint8_t *rgba8,
                          size_t stride) {
    if (!c || !rgba8) return VELLO_HIP_E_INVALID;
    if ((uint64_t)x + width > c->atlas_w || (uint64_t)y + height > c->atlas_h) {
        c->last_error = "write_image outside the atlas";
        return VELLO_HIP_E_INVALID;
    }
    if (width == 0 || height == 0) return VELLO_HIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (stride == 0) stride = (size_t)width * 4u;
    int r = ensure_upload_stream(c);
    if (r) return r;
    // the caller owns the pixels only for the duration of the call: into pinned memory now, DMA later
    const size_t row_bytes = (size_t)width * 4u;
    Staging *st = nullptr;
    if ((r = acquire_staging(c, row_bytes * height, st))) return r;
    for (uint32_t row = 0; row < height; row++) std::memcpy((char *)st->host + row * row_bytes, rgba8 + row * stride, row_bytes);
    // frames already enqueued may sample the texels this upload replaces: it runs behind all of them
    for (auto &l : c->lanes) {
        if (!l.stream || !l.used) continue;
        HIP_TRY(c, hipEventRecord(c->lane_mark, l.stream));
        HIP_TRY(c, hipStreamWaitEvent(c->upload_stream, c->lane_mark, 0));
    }
    HIP_TRY(c, hipMemcpy2DAsync((char *)c->atlas.ptr + ((size_t)y * c->atlas_w + x) * 4u, (size_t)c->atlas_w * 4u, st->host, row_bytes,
                                row_bytes, height, hipMemcpyHostToDevice, c->upload_stream));
    HIP_TRY(c, hipEventRecord(st->done, c->upload_stream));
    st->busy = true;
    // ... and every frame enqueued from here on runs behind it (prepare_frame)
    HIP_TRY(c, hipEventRecord(c->atlas_ready, c->upload_stream));
    c->atlas_epoch += 1u;
    return VELLO_HIP_OK;
}

int vello_hip_render_resident(vello_hip_ctx *c, const vello_hip_render_params *params, void *out_device, size_t out_stride) {
    if (!c) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t li = c->next_lane % c->n_active;
    Lane &l = c->lanes[li];
    int r;
    if (l.use_own) {
        // the lane last rendered a vello_hip_render_frame scene: resident frames always show the scene of
        // vello_hip_upload_scene, whichever lane the rotation has reached
        if (!c->shared.resident) {
            c->last_error = "no scene uploaded (vello_hip_render_frame scenes are private to their frame)";
            return VELLO_HIP_E_INVALID;
        }
        HIP_TRY(c, hipStreamSynchronize(l.stream));
        l.use_own = false;
        if ((r = alloc_lane_scene(c, l, c->shared))) return r;
    }
    c->next_lane = (li + 1u) % c->n_active;
    c->last_lane = li;
    Frame f;
    r = prepare_frame(c, l, params, out_device, out_stride, f, false);
    if (r) return r;
    int last = VELLO_HIP_STAGE_FINE;
    if (const uint32_t ls = (c->debug_flags >> VELLO_HIP_DEBUG_LAST_STAGE_SHIFT) & 15u) last = (int)ls - 1 < last ? (int)ls - 1 : last;  // (measurement seam)
    return run_stage_range(c, l, f, 0, last);
}

int vello_hip_run_stages(vello_hip_ctx *c, const vello_hip_render_params *params, int first, int last) {
    if (!c || first < 0 || last >= VELLO_HIP_STAGE_COUNT || first > last) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    Lane &l = c->lanes[c->last_lane];
    Frame f;
    int r = prepare_frame(c, l, params, nullptr, 0, f, true);
    if (r) return r;
    if ((r = run_stage_range(c, l, f, first, last))) return r;
    HIP_TRY(c, hipStreamSynchronize(l.stream));
    return VELLO_HIP_OK;
}

int vello_hip_get_bump(vello_hip_ctx *c, vello_hip_bump *out) {
    if (!c || !out || !c->lanes[c->last_lane].zero_region.ptr) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    Lane &l = c->lanes[c->last_lane];
    HIP_TRY(c, hipStreamSynchronize(l.stream));
    HIP_TRY(c, hipMemcpy(out, l.zero_region.ptr, sizeof(vello_hip_bump), hipMemcpyDeviceToHost));
    return VELLO_HIP_OK;
}

int vello_hip_sync_frame(vello_hip_ctx *c, uint32_t age) {
    if (!c || age >= c->n_active) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t n = c->n_active;
    Lane &l = c->lanes[(c->last_lane + n - age) % n];
    HIP_TRY(c, hipStreamSynchronize(l.stream));
    // (once per scene: what flatten counted, so that later frames can leave out a launch that would exit at once)
    SceneSlot &sc = slot_of(c, l);
    if (l.used && l.flatten_ran && l.zero_region.ptr && l.frame_generation == sc.generation && (sc.stroke_lines < 0 || (l.slices_on && sc.slice_demand < 0))) {
        Control ctl;
        HIP_TRY(c, hipMemcpy(&ctl, l.zero_region.ptr, sizeof ctl, hipMemcpyDeviceToHost));
        if (ctl.bump.failed == 0u) {
            sc.stroke_lines = (int64_t)ctl.heavy_count[2];
            sc.heavy_curves = (int64_t)ctl.heavy_count[0];
            sc.heavy_strokes = (int64_t)ctl.heavy_count[1];
            sc.soup_lines = (int64_t)ctl.bump.lines;
            if (l.slices_on && (int64_t)ctl.slice_items > sc.slice_demand) sc.slice_demand = (int64_t)ctl.slice_items;
        }
    }
    return VELLO_HIP_OK;
}

// bump.failed of ONE lane's latest frame -> error code.  Failures are per frame: vello_hip_render judges the lane that
// rendered its frame, vello_hip_sync the lanes of the current rotation, and vello_hip_grow_pools voids the frames that
// overflowed the old pools (l.used) -- so a retry on another lane is not failed by the lane that overflowed.
static int check_lane(vello_hip_ctx *c, Lane &l) {
    if (!l.used || !l.zero_region.ptr) return VELLO_HIP_OK;
    Control ctl;
    HIP_TRY(c, hipMemcpy(&ctl, l.zero_region.ptr, sizeof ctl, hipMemcpyDeviceToHost));
    vello_hip_bump b;
    std::memcpy(&b, &ctl.bump, sizeof b);
    if (b.failed == 0u) {
        SceneSlot &sc = slot_of(c, l);
        if (l.flatten_ran && l.frame_generation == sc.generation) {  // (flatten ran to its end)
            sc.stroke_lines = (int64_t)ctl.heavy_count[2];
            sc.heavy_curves = (int64_t)ctl.heavy_count[0];
            sc.heavy_strokes = (int64_t)ctl.heavy_count[1];
            sc.soup_lines = (int64_t)ctl.bump.lines;
            if (l.slices_on && (int64_t)ctl.slice_items > sc.slice_demand) sc.slice_demand = (int64_t)ctl.slice_items;
        }
        return VELLO_HIP_OK;
    }
    if ((b.failed & FAILED_SCENE) != 0u) {
        c->last_error = "the path tag stream needs more path data, transforms or styles than the scene buffer holds";
        return VELLO_HIP_E_INVALID;
    }
    if ((b.failed & FAILED_INTERNAL) != 0u) {
        // a spin bound tripped (lookback.h, k_front's barrier): not a pool overflow -- growing pools would not help.  The lane is idle
        // here (the caller waited for it): its barrier counter starts over.
        reset_front_sync(l, l.stream);
        (void)hipStreamSynchronize(l.stream);
        c->last_error = "an engine-internal wait gave up (bump.failed bit 31): the frame is discarded";
        return VELLO_HIP_E_INTERNAL;
    }
    char msg[160];
    std::snprintf(msg, sizeof msg, "bump.failed=0x%x (lines %u, binning %u, tile %u, seg_counts %u, segments %u, ptcl %u)", b.failed,
                  b.lines, b.binning, b.tile, b.seg_counts, b.segments, b.ptcl);
    c->last_error = msg;
    return VELLO_HIP_E_CAPACITY;
}

int vello_hip_sync(vello_hip_ctx *c) {
    if (!c) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = sync_all(c);
    if (r) return r;
    if (!c->have_cfg) return VELLO_HIP_OK;
    // only the lanes of the current rotation: a lane dropped by set_frames_in_flight keeps its old control block
    int first = VELLO_HIP_OK;
    for (uint32_t i = 0; i < c->n_active && i < c->lanes.size(); i++) {
        r = check_lane(c, c->lanes[i]);
        if (r == VELLO_HIP_E_HIP) return r;
        if (r && !first) first = r;
    }
    return first;
}

uint32_t vello_hip_last_render_attempts(vello_hip_ctx *c) { return c ? c->last_render_attempts : 0u; }
uint64_t vello_hip_fused_launches(vello_hip_ctx *c) { return c ? c->fused_launches : 0u; }

void *vello_hip_get_stream(vello_hip_ctx *c) { return c ? (void *)c->lanes[c->last_lane].stream : nullptr; }

int vello_hip_get_capacities(vello_hip_ctx *c, vello_hip_capacities *out) {
    if (!c || !out) return VELLO_HIP_E_INVALID;
    *out = c->caps;
    return VELLO_HIP_OK;
}

// The step the reference leaves as a TODO (lib.rs:753-764, "apply logic to determine whether we need to rerun
// coarse"): size every pool whose counter exceeded it for the reported demand.  A stage that overflows stops the
// stages behind it (bump.failed, shared/bump.wgsl:5-9), so a frame may need one call per overflowing stage.
int vello_hip_grow_pools(vello_hip_ctx *c, const vello_hip_bump *demand, vello_hip_capacities *new_caps) {
    if (!c || !demand) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    vello_hip_capacities d = c->caps;
    bool grew = false;
    // 25 % headroom, rounded up to 64 Ki elements; counters are u32, so is every pool
    auto want = [&](uint32_t &cap, uint64_t need) {
        if (need <= cap) return;
        uint64_t n = need + need / 4u;
        n = (n + 0xffffu) & ~(uint64_t)0xffffu;
        if (n > 0xffff0000ull) n = 0xffff0000ull;
        if (n > cap) {
            cap = (uint32_t)n;
            grew = true;
        }
    };
    want(d.lines, demand->lines);
    want(d.bin_data, (uint64_t)demand->binning + slot_of(c, c->lanes[c->last_lane]).layout.bin_data_start);
    want(d.tiles, demand->tile);
    want(d.seg_counts, demand->seg_counts);
    want(d.segments, demand->segments);
    want(d.blend_spill, demand->blend);
    if (c->have_cfg) {
        uint64_t dyn_start = (uint64_t)c->cfg.width_in_tiles * c->cfg.height_in_tiles * PTCL_INITIAL_ALLOC;
        want(d.ptcl, dyn_start + demand->ptcl + PTCL_INCREMENT);
    }
    if (d.segments < d.seg_counts) {
        d.segments = d.seg_counts;
        grew = true;
    }
    if (new_caps) *new_caps = d;
    if (!grew) {
        c->last_error = "grow_pools: no counter exceeds its pool";
        return VELLO_HIP_E_INVALID;
    }
    int r = sync_all(c);
    if (r) return r;
    if ((r = commit_caps(c, d))) return r;
    for (auto &l : c->lanes) l.used = false;  // the frames rendered into the old pools are void; their failure has been acted upon
    return VELLO_HIP_OK;
}

// The one exchange step of the path for a host that owns one context per GPU in ONE process: every
// context's finished frame goes to `dst_device` with hipMemcpyPeerAsync on that context's own copy stream -- the SDMA
// engines move it over the peer's own xGMI link, no CU is involved and the copies of different peers run concurrently
// -- ordered behind the frame the context enqueued last by an event, not by a host wait.  (One process per GPU, as
// bench.py runs, gathers with RCCL instead: vello_amd/distributed.py.)
int vello_hip_gather_frames(vello_hip_ctx *const *ctxs, uint32_t n, int dst_device, const void *const *src_frames, void *const *dst_frames,
                            size_t frame_bytes) {
    if (!ctxs || !src_frames || !dst_frames || n == 0) return VELLO_HIP_E_INVALID;
    for (uint32_t i = 0; i < n; i++) {
        vello_hip_ctx *c = ctxs[i];
        if (!c || !src_frames[i] || !dst_frames[i]) return VELLO_HIP_E_INVALID;
        HIP_TRY(c, hipSetDevice(c->device));
        if (!c->copy_stream) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&c->frame_done, hipEventDisableTiming));
            if (c->device != dst_device) {
                hipError_t e = hipDeviceEnablePeerAccess(dst_device, 0);  // direct xGMI path; already enabled is fine
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // staged copies still work
            }
        }
        HIP_TRY(c, hipEventRecord(c->frame_done, c->lanes[c->last_lane].stream));
        HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->frame_done, 0));
        HIP_TRY(c, hipMemcpyPeerAsync(dst_frames[i], dst_device, src_frames[i], c->device, frame_bytes, c->copy_stream));
    }
    return VELLO_HIP_OK;
}

int vello_hip_gather_wait(vello_hip_ctx *const *ctxs, uint32_t n) {
    if (!ctxs) return VELLO_HIP_E_INVALID;
    for (uint32_t i = 0; i < n; i++) {
        vello_hip_ctx *c = ctxs[i];
        if (!c) return VELLO_HIP_E_INVALID;
        if (!c->copy_stream) continue;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    }
    return VELLO_HIP_OK;
}

int vello_hip_set_debug_flags(vello_hip_ctx *c, uint32_t flags) {
    if (!c) return VELLO_HIP_E_INVALID;
    if ((c->debug_flags ^ flags) & VELLO_HIP_DEBUG_FINE_SLICES) {  // what earlier frames asked for says nothing about the other slice size
        c->shared.slice_demand = -1;
        for (auto &l : c->lanes) l.own.slice_demand = -1;
    }
    c->debug_flags = flags;
    return VELLO_HIP_OK;
}

int vello_hip_set_auto_grow(vello_hip_ctx *c, int enabled) {
    if (!c) return VELLO_HIP_E_INVALID;
    c->auto_grow = enabled != 0;
    return VELLO_HIP_OK;
}

// Robust mode, before the first attempt: when the scene is large against the current pools, size them from
// vello_hip_estimate_capacities (estimate.rs's counting rules on the packed scene) instead of finding the demand by
// overflowing stage after stage.  The cheap test first: every path segment yields at least one line.
static int presize_pools(vello_hip_ctx *c, const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                         const vello_hip_render_params *params) {
    if (!scene || !layout || !params || layout->path_tag_base > layout->path_data_base || (size_t)layout->path_data_base * 4u > scene_len)
        return VELLO_HIP_OK;  // the upload refuses it with a message
    const uint8_t *tags = scene + (size_t)layout->path_tag_base * 4u;
    const size_t n_tags = ((size_t)layout->path_data_base - layout->path_tag_base) * 4u;
    uint64_t n_seg = 0;
    for (size_t i = 0; i < n_tags; i++) n_seg += (tags[i] & 3u) != 0u;
    const uint64_t n_tiles = (uint64_t)((params->width + 15u) / 16u) * ((params->height + 15u) / 16u);
    if (4u * n_seg <= c->caps.lines && 8u * n_seg <= c->caps.seg_counts && 4u * n_seg + 8u * layout->n_paths <= c->caps.tiles &&
        160u * n_tiles + 16u * n_seg <= c->caps.ptcl)
        return VELLO_HIP_OK;
    vello_hip_capacities est;
    if (vello_hip_estimate_capacities(scene, scene_len, layout, params, &est) != VELLO_HIP_OK) return VELLO_HIP_OK;
    vello_hip_capacities d = c->caps;
    bool grew = false;
    auto want = [&](uint32_t &cap, uint32_t need) {
        if (need > cap) {
            cap = need;
            grew = true;
        }
    };
    want(d.lines, est.lines);
    want(d.bin_data, est.bin_data);
    want(d.tiles, est.tiles);
    want(d.seg_counts, est.seg_counts);
    want(d.segments, est.segments);
    want(d.ptcl, est.ptcl);
    if (d.segments < d.seg_counts) d.segments = d.seg_counts;
    if (!grew) return VELLO_HIP_OK;
    // The estimate comes from caller-supplied geometry (up to 0xffff0000 elements per pool): one that the device cannot
    // hold is not an error of this frame -- the pools stay and the overflow-and-grow loop of vello_hip_render finds the
    // real demand, which is usually far below the estimator's bound.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return VELLO_HIP_OK;
    const size_t n_lanes = c->lanes.size(), held = pool_bytes(c->caps, c->aa_mask) * n_lanes, asked = pool_bytes(d, c->aa_mask) * n_lanes;
    if (asked > held && asked - held > free_b - free_b / 8u) return VELLO_HIP_OK;
    int r = sync_all(c);
    if (r) return r;
    if (commit_caps(c, d)) c->last_error.clear();  // capacities unchanged (commit_caps), fall back to the loop
    return VELLO_HIP_OK;
}

int vello_hip_render(vello_hip_ctx *c, const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                     const vello_hip_render_params *params, const uint32_t *ramps, uint32_t n_ramps, void *out_rgba8, size_t out_stride,
                     int out_is_device, vello_hip_bump *bump_out) {
    if (!c || !params) return VELLO_HIP_E_INVALID;
    int r;
    if (c->auto_grow && (r = presize_pools(c, scene, scene_len, layout, params))) return r;
    r = vello_hip_upload_scene(c, scene, scene_len, layout, ramps, n_ramps);
    if (r) return r;
    int sync_r = VELLO_HIP_OK;
    // robust mode: re-run with grown pools until the frame fits (each overflowing stage hides the demand of
    // the stages behind it, so a handful of rounds at most)
    for (int attempt = 0; attempt < 8; attempt++) {
        c->last_render_attempts = (uint32_t)attempt + 1u;
        r = vello_hip_render_resident(c, params, out_is_device ? out_rgba8 : nullptr, out_stride);
        if (r) return r;
        Lane &l = c->lanes[c->last_lane];
        HIP_TRY(c, hipStreamSynchronize(l.stream));
        vello_hip_bump b;
        HIP_TRY(c, hipMemcpy(&b, l.zero_region.ptr, sizeof b, hipMemcpyDeviceToHost));
        sync_r = check_lane(c, l);  // this frame's lane only: another lane's older failure is not this frame's
        if (bump_out) *bump_out = b;
        if (sync_r != VELLO_HIP_E_CAPACITY || !c->auto_grow) break;
        if (vello_hip_grow_pools(c, &b, nullptr) != VELLO_HIP_OK) break;
    }
    if (sync_r) return sync_r;
    if (out_rgba8 && !out_is_device) {
        size_t row = (size_t)params->width * 4u;
        size_t stride = out_stride ? out_stride : row;
        HIP_TRY(c, hipMemcpy2D(out_rgba8, stride, c->lanes[c->last_lane].buf[VELLO_HIP_BUF_OUTPUT].ptr, row, row, params->height,
                               hipMemcpyDeviceToHost));
    }
    return VELLO_HIP_OK;
}

size_t vello_hip_buffer_size(vello_hip_ctx *c, int id) {
    if (!c || id < 0 || id >= VELLO_HIP_BUF_COUNT) return 0;
    return find_buf(c, id)->size;
}

int vello_hip_read_buffer(vello_hip_ctx *c, int id, void *dst, size_t offset, size_t size) {
    if (!c || id < 0 || id >= VELLO_HIP_BUF_COUNT || !dst) return VELLO_HIP_E_INVALID;
    DevBuf *b = find_buf(c, id);
    if (!b->ptr || offset + size > b->size) {
        c->last_error = "read_buffer out of range";
        return VELLO_HIP_E_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int r = sync_all(c);
    if (r) return r;
    HIP_TRY(c, hipMemcpy(dst, (const char *)b->ptr + offset, size, hipMemcpyDeviceToHost));
    return VELLO_HIP_OK;
}

int vello_hip_write_buffer(vello_hip_ctx *c, int id, const void *src, size_t offset, size_t size) {
    if (!c || id < 0 || id >= VELLO_HIP_BUF_COUNT || !src) return VELLO_HIP_E_INVALID;
    DevBuf *b = find_buf(c, id);
    if (!b->ptr || offset + size > b->size) {
        c->last_error = "write_buffer out of range";
        return VELLO_HIP_E_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int r = sync_all(c);
    if (r) return r;
    HIP_TRY(c, hipMemcpy((char *)b->ptr + offset, src, size, hipMemcpyHostToDevice));
    return VELLO_HIP_OK;
}

int vello_hip_set_profiling(vello_hip_ctx *c, uint32_t stage_mask) {
    if (!c) return VELLO_HIP_E_INVALID;
    c->prof_mask = stage_mask;
    return VELLO_HIP_OK;
}

int vello_hip_get_stage_ms(vello_hip_ctx *c, float ms_out[VELLO_HIP_STAGE_COUNT], uint32_t count_out[VELLO_HIP_STAGE_COUNT]) {
    if (!c) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = drain_events(c);
    if (r) return r;
    for (int i = 0; i < VELLO_HIP_STAGE_COUNT; i++) {
        if (ms_out) ms_out[i] = c->stage_ms[i];
        if (count_out) count_out[i] = c->stage_count[i];
        c->stage_ms[i] = 0.f;
        c->stage_count[i] = 0;
    }
    return VELLO_HIP_OK;
}

int vello_hip_get_kernel_ms(vello_hip_ctx *c, int stage, float ms_out[3], uint32_t *count_out) {
    if (!c || stage < 0 || stage >= VELLO_HIP_STAGE_COUNT || !ms_out) return VELLO_HIP_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = drain_events(c);
    if (r) return r;
    for (int k = 0; k < 3; k++) {
        ms_out[k] = c->kernel_ms[stage][k];
        c->kernel_ms[stage][k] = 0.f;
    }
    if (count_out) *count_out = c->kernel_count[stage];
    c->kernel_count[stage] = 0;
    return VELLO_HIP_OK;
}

}  // extern "C"
