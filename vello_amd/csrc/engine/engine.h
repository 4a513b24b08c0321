// Engine-internal interface between the host driver (engine.hip) and the kernel translation units.
#pragma once
#include <cstddef>
#include "common.h"

namespace vk {

// Single-pass decoupled look-back state: per partition, NF granules for the workgroup aggregate
// followed by NF granules for the inclusive prefix.  A granule is one naturally aligned 8-byte
// {status:32, value:32} word written exactly once per frame by ONE relaxed agent-scope store
// (write-through), so it is never torn and needs no separate flag or fence
// (cdna_hip_programming.md guideline 16, form R2).
constexpr uint32_t SCAN_STATUS_AGG = 1, SCAN_STATUS_PREFIX = 2;
constexpr uint32_t PATHTAG_PART_WORDS = 1024;  // 256 threads x 4 tag words (= 4096 tags)
constexpr uint32_t DRAW_PART = 256;            // draw objects per partition
#ifndef VK_FLATTEN_TPT
#define VK_FLATTEN_TPT 4
#endif
constexpr uint32_t FLATTEN_TAGS_PER_THREAD = VK_FLATTEN_TPT;
constexpr uint32_t FLATTEN_BLOCK_TAGS = 256 * FLATTEN_TAGS_PER_THREAD;
#ifndef VK_PC_LPT
#define VK_PC_LPT 4
#endif
constexpr uint32_t PATH_COUNT_LINES_PER_THREAD = VK_PC_LPT;
constexpr uint32_t PATH_COUNT_CHUNK = 256 * PATH_COUNT_LINES_PER_THREAD;
// A soup of fewer lines than this (known from a finished frame of the scene) is cut into chunks of 256 lines instead of 1 024:
// a chunk is 15-20 us of dependent phases whatever its size, and the tiger's 15 000 lines are 15 workgroups of the large kind
// on 256 CUs.  (768 workgroups of k_path_count are resident at once.)
constexpr int64_t PATH_COUNT_SMALL_MAX_LINES = 768 * 256;
// Spin bound for look-back waits: a predecessor always holds a smaller ticket, so it is resident
// or finished; the bound only turns a driver-level hang into a reported failure.
constexpr uint32_t SPIN_LIMIT = 1u << 24;
constexpr uint32_t FAILED_INTERNAL = 0x80000000u;  // set in bump.failed when a spin bound trips
// set in bump.failed by the pathtag scan when the tag stream asks for more path data, transforms, styles or paths
// than the scene buffer / layout hold (WebGPU's robust buffer access makes this harmless upstream; HIP has none):
// every later stage that would index with those counts bails out, the frame reports VELLO_HIP_E_INVALID
constexpr uint32_t FAILED_SCENE = 0x40000000u;
constexpr uint32_t FINE_WORK_BUCKETS = 32;  // buckets of 32 command words (the last one: 992 and more)
// Stroked lines get workgroups of their own (beside the heavy list's in k_flatten_main with one frame in flight, as
// k_flatten_strokes ahead of k_flatten_heavy with several) once they alone fill the chip twice over at 12 waves per CU
// (256 CUs x 12 x 64 lanes); below that they are entries of the heavy list, whose duration the curves set anyway.
constexpr uint32_t FLATTEN_STROKE_KERNEL_MIN_LINES = 2u * 256u * 12u * 64u;
// command words from which a tile counts as long: its wave raises its issue priority (s_setprio) in k_fine
constexpr uint32_t FINE_HEAVY_WORDS = 384;
// fine, MSAA modes: a tile whose list holds >= FINE_SLICE_MIN_FILLS FILLs is cut into slices of FINE_SLICE_FILLS fills; every
// slice is a work item of its own (one wave computes the coverage of its fills into the coverage scratch), and the wave that
// finishes a tile's last slice composites the tile from the scratch (fine.hip).  k_fine's launch is as long as its longest
// tile's chain of fills otherwise (d2: 137 fills, 258 us against 117 us of balanced work).
#ifndef VK_SLICE_FILLS
#define VK_SLICE_FILLS 32
#endif
#ifndef VK_SLICE_MIN_FILLS
#define VK_SLICE_MIN_FILLS 96
#endif
constexpr uint32_t FINE_SLICE_FILLS = VK_SLICE_FILLS, FINE_SLICE_MIN_FILLS = VK_SLICE_MIN_FILLS;
// With frames in flight the tail of k_fine's launch is filled by the other frames' kernels, and what slicing costs -- the coverage
// through memory, a second pass over the list -- is no longer paid back: the threshold is higher there (round 6,
// profiles/r06_ab_slices_in_flight.txt).
#ifndef VK_SLICE_MIN_FILLS_IN_FLIGHT
#define VK_SLICE_MIN_FILLS_IN_FLIGHT 192
#endif
#ifndef VK_SLICE_FILLS_IN_FLIGHT
#define VK_SLICE_FILLS_IN_FLIGHT VK_SLICE_FILLS
#endif
constexpr uint32_t FINE_SLICE_FILLS_IN_FLIGHT = VK_SLICE_FILLS_IN_FLIGHT, FINE_SLICE_MIN_FILLS_IN_FLIGHT = VK_SLICE_MIN_FILLS_IN_FLIGHT;
constexpr uint32_t FINE_SLICE_FILLS_FORCED = 4, FINE_SLICE_MIN_FILLS_FORCED = 5;  // VELLO_HIP_DEBUG_FINE_SLICES
struct SliceItem {
    uint32_t tile_ix;     // ~0: a hole left by a tile whose slices did not fit the capacity
    uint32_t k_and_n;     // slice | slices of the tile << 16
    uint32_t cov_base;    // first word of the tile's coverage scratch: 64 words (one byte per pixel) per FILL
    uint32_t first_item;  // index of the tile's first item: its arrival counter is slice_counters[first_item]
};
// clip matching (clip.hip): clips per partition (a thread each) and the most partitions the one-workgroup stack pass holds
// in LDS; beyond CLIP_PART * CLIP_MAX_PARTS clips the one-wave stack machine (draw.hip) runs instead
constexpr uint32_t CLIP_PART = 256;
constexpr uint32_t CLIP_MAX_PARTS = 2048;
struct ClipEl { uint32_t clip_ix; float x0, y0, x1, y1; };  // an open BeginClip: its index in clip_inp, its box ∩ its local ancestors'
__host__ __device__ inline uint32_t clip_parts_pad(uint32_t parts) {  // leaves of the min-tree over the partitions
    uint32_t n = 2u;
    while (n < parts) n <<= 1;
    return n;
}
// words of Frame::clip_stack: the sequential machine's spill area (6 words per clip) or the partitions' scratch
// (open pushes, Bic, height, box below: 1287 words per partition; then the min-tree)
inline size_t clip_scratch_words(uint32_t n_clips) {
    size_t parts = (n_clips + CLIP_PART - 1u) / CLIP_PART;
    size_t par = parts * (CLIP_PART * 5u + 4u + 2u + 1u) + 2u * (size_t)clip_parts_pad((uint32_t)parts);
    size_t seq = ((size_t)n_clips + 1u) * 6u;
    return par > seq ? par : seq;
}

// Words of the per-frame control block (zeroed by ONE hipMemsetAsync per frame, together with
// the bump allocators and both look-back state arrays which follow it in the same allocation).
struct Control {
    Bump bump;              // must be first: VELLO_HIP_BUF_BUMP aliases it
    uint32_t ticket_pathtag;
    uint32_t ticket_draw;
    uint32_t heavy_count[4];  // flatten: tags queued by k_flatten_light: [0] fill curves, [1] strokes, [2] stroked lines; [3] stroked lines
                              // the stroke workgroups of k_flatten_main hand on to k_flatten_tail
    uint32_t pad[2];
    uint32_t work_count[FINE_WORK_BUCKETS];  // coarse -> fine: tiles per bucket of command-list length (k_fine runs the long ones first)
    uint32_t slice_items;   // coarse -> fine: SliceItems handed out (may run past the capacity: fine clamps); work_count[BUCKETS]
    uint32_t cov_words;     // coarse -> fine: words of the coverage scratch handed out
    uint32_t pad2[48 - FINE_WORK_BUCKETS - 2];
    // flatten: arcs the stroke workgroups set aside, counted in FLATTEN_ARC_SHARDS sub-lists (workgroup b appends to shard b mod
    // 64): ONE counter took 2 800 same-address atomics of ~12 ns each in a burst -- 30 us on the road-map scene
    uint32_t arc_count[64];
};
static_assert(sizeof(Control) == 512, "Control");
// (coarse and fine reach slice_items / cov_words through the work_count pointer)
static_assert(offsetof(Control, slice_items) == offsetof(Control, work_count) + 4u * FINE_WORK_BUCKETS, "Control::slice_items follows work_count");
static_assert(offsetof(Control, cov_words) == offsetof(Control, slice_items) + 4u, "Control::cov_words follows slice_items");
constexpr uint32_t FLATTEN_ARC_SHARDS = 64;
// Stroke workgroups of a scene of at most n_seg_max segments.  Side by side with the heavy list's workgroups (one frame in flight,
// k_flatten_main): one per round of 256 stroked lines, up to 4 096.  As a launch of their own (frames in flight,
// k_flatten_strokes): 512 -- the two a CU holds -- striding over the rounds: four frames in flight +1.9 % on the road map; the same
// grid for k_flatten_main is 2.8 % slower one frame at a time (profiles/r04_ab_s21_strokes_grid.txt).
#ifndef VK_STROKES_GRID_SIDE_BY_SIDE
#define VK_STROKES_GRID_SIDE_BY_SIDE 4096u
#endif
#ifndef VK_STROKES_GRID_OWN_LAUNCH
#define VK_STROKES_GRID_OWN_LAUNCH 384u  // (scripts/emu_variant_check.sh sets both to 2: several rounds per workgroup on the emulator's small scenes)
#endif
inline uint32_t flatten_strokes_grid(uint32_t n_seg_max, bool side_by_side) {
    const uint32_t cap = side_by_side ? VK_STROKES_GRID_SIDE_BY_SIDE : VK_STROKES_GRID_OWN_LAUNCH;
    const uint32_t g = (n_seg_max + 255u) / 256u;
    return g > cap ? cap : (g < 1u ? 1u : g);
}
// the arcs one shard of the arc list can be given (256 per round of each of its workgroups); the list holds FLATTEN_ARC_SHARDS x
// that many 64-byte items
inline uint32_t flatten_arc_shard_cap(uint32_t n_seg_max, bool side_by_side) {
    const uint32_t grid = flatten_strokes_grid(n_seg_max, side_by_side);
    const uint32_t rounds = ((n_seg_max + 255u) / 256u + grid - 1u) / grid;
    return ((grid + FLATTEN_ARC_SHARDS - 1u) / FLATTEN_ARC_SHARDS) * (rounds < 1u ? 1u : rounds) * 256u;
}

struct Frame {
    Config cfg;  // host copy; kernels receive it by value
    uint32_t n_tag_words;
    uint32_t n_scene_words;  // length of the packed scene
    uint32_t aa;
    // device pointers
    const uint32_t *scene;
    Control *control;
    uint32_t *heavy_list;   // flatten: tag indices that need the Euler-spiral / stroker path
    uint32_t *arc_items;    // flatten: 16 words per round join / cap arc that a stroke workgroup leaves to the heavy code
    unsigned long long *pathtag_state;  // [n_pathtag_parts][2][5]
    unsigned long long *draw_state;     // [n_draw_parts][2][4]
    TagMonoid *tag_monoids;
    PathBbox *path_bboxes;
    LineSoup *lines;
    DrawMonoid *draw_monoids;
    uint32_t *info_bin_data;
    Clip *clip_inp;
    Bbox4 *clip_bboxes;
    Bbox4 *draw_bboxes;
    BinHeader *bin_headers;
    Path *paths;
    Tile *tiles;
    SegmentCount *seg_counts;
    Segment *segments;
    uint32_t *ptcl;
    uint32_t *blend_spill;
    CoarseEl *coarse_el;     // coarse: one record per draw object (k_coarse_prep)
    uint32_t *tile_bits;     // coarse: three bits per tile of the pool (segments present / backdrop zero / backdrop even), a word per 8 tiles
    uint32_t *tile_order;    // coarse -> fine: [bucket][n_tiles] tile indices, filled up to control->work_count[bucket]
    SliceItem *slice_items;  // coarse -> fine: the slices of the long tiles (MSAA modes), slice_cap entries
    uint32_t *slice_counters;  // per first item: slices of the tile that have finished
    uint32_t *cov;           // fine: coverage scratch of the sliced tiles, cov_cap words
    uint32_t slice_cap, cov_cap;
    uint32_t slice_fills, slice_min_fills;  // 0 / 0: no slicing (area AA)
    uint32_t *clip_stack;  // clip_scratch_words(n_clips): scratch of the partitioned clip kernels / spill area of the sequential one
    uint8_t *output;
    size_t out_stride;
    const uint32_t *ramps;
    uint32_t n_ramps;
    const uint32_t *atlas;  // RGBA8 image atlas (render.rs:160-203), atlas_w x atlas_h texels
    uint32_t atlas_w, atlas_h;
    uint32_t stroke_kernel_min_lines;  // flatten: stroked lines from which stroke workgroups take them (FLATTEN_STROKE_KERNEL_MIN_LINES; 0 with VELLO_HIP_DEBUG_STROKE_KERNEL)
    bool path_count_small;  // path_count: chunks of 256 lines (PATH_COUNT_SMALL_MAX_LINES)
    bool flatten_side_by_side;  // flatten: stroke workgroups in the heavy list's launch (one frame in flight) instead of a kernel before it
    bool launch_stroke_kernel;  // false when an earlier frame of the same scene showed that stroke workgroups would exit at once
    bool flatten_coop;          // flatten's heavy list by the kernels of the wave-cooperative walk (flatten_walk.inc) instead of round 4's
    bool sequential_clip;  // VELLO_HIP_DEBUG_SEQ_CLIP: the one-wave stack machine whatever the clip count
    bool no_cull;  // VELLO_HIP_DEBUG_NO_CULL: coarse emits every draw, as the reference does (exact PTCL / segment diffs)
    bool brushes;  // the scene has gradient / image / blurred-rect draw objects (selects fine's specialisation)
    const uint32_t *mask_lut8;
    const uint32_t *mask_lut16;
    uint32_t zero_bytes;     // the lane's zero region: Control + both look-back states (k_front clears it itself)
    uint32_t *front_sync;    // k_front's grid-barrier counter (per lane; only ever grows)
    Bump *bump() const { return &control->bump; }
};

void launch_pathtag_scan(const Frame &f, hipStream_t s);
// (mid: when not null, an event is recorded behind every kernel of the stage but the last: per-KERNEL times of a stage of
// several kernels, vello_hip_get_kernel_ms)
// with_draw_scan: the draw stage's workgroups ride in k_flatten_light's launch (the caller then leaves launch_draw_scan out)
// light_done: k_front has run the light pass (and the draw stage's workgroups) already
void launch_flatten(const Frame &f, hipStream_t s, hipEvent_t *mid = nullptr, bool with_draw_scan = false, bool light_done = false);
// Small scenes: the workgroups of consecutive stages as ONE launch (k_front, flatten.hip).  `stages`: FRONT_* bits, consecutive stages;
// returns what the launch adds to *f.front_sync (the caller keeps the counter's value: sync_base is the value before the launch).
constexpr uint32_t FRONT_ZERO = 1u, FRONT_PATHTAG = 2u, FRONT_LIGHT = 4u, FRONT_HEAVY = 8u, FRONT_BINNING = 16u, FRONT_TILE_ALLOC = 32u;
// (a turn per workgroup and stage at most -- FRONT_MAX_WG, flatten.hip: 16 light-pass blocks of 1 024 tags, 16 blocks of 256 draw objects; a
// scene of 2 000 paths with twice these is 1 % slower one frame at a time and 4 % with four in flight when its stages share launches,
// the scenes below gain 1-10 % and 3-50 %: profiles/r05_small_scene_latency.jsonl)
constexpr uint32_t FRONT_MAX_TAGS = 16384u, FRONT_MAX_DRAW_OBJECTS = 4096u;
constexpr uint32_t FRONT_TINY_SEGMENTS = 64u;  // up to here the heavy list joins the launch, which is then ONE workgroup
uint32_t launch_front(const Frame &f, hipStream_t s, uint32_t stages, bool with_draw_scan, uint32_t sync_base);
inline uint32_t flatten_n_seg_max(const Frame &f) {
    // (a segment owns at least one word of path data, so the path-data stream bounds the segments even though the tag stream is padded)
    const uint32_t n_tags = f.n_tag_words * 4u, n_data = f.cfg.layout.draw_tag_base - f.cfg.layout.path_data_base;
    return n_tags < n_data ? n_tags : n_data;
}
void launch_draw_scan(const Frame &f, hipStream_t s);
void launch_clip(const Frame &f, hipStream_t s);             // clip.hip
void launch_clip_sequential(const Frame &f, hipStream_t s);  // draw.hip
void launch_binning(const Frame &f, hipStream_t s);
void launch_tile_alloc(const Frame &f, hipStream_t s);
void launch_binning_tile_alloc(const Frame &f, hipStream_t s);
void launch_path_count(const Frame &f, hipStream_t s);
void launch_backdrop(const Frame &f, hipStream_t s);
void launch_coarse(const Frame &f, hipStream_t s, hipEvent_t *mid = nullptr);
int enable_coarse_lds();  // hipError_t of the per-device dynamic-LDS opt-in
void launch_path_tiling(const Frame &f, hipStream_t s);
void launch_fine(const Frame &f, hipStream_t s);

}  // namespace vk
