// coarse: per-tile command lists (PTCL) in draw order.
// Reference: vello_shaders/shader/coarse.wgsl:62-471 (vello/src/render.rs:470-484), PTCL format
// shared/ptcl.wgsl:6-111; CPU twin cpu/coarse.rs.
//
// The reference runs one workgroup per 16x16-tile bin, one thread per tile; each batch of 256 draw objects of the
// bin is merged from the bin lists, a coverage pass reads the Tile record of EVERY (draw object, tile-in-bbox) pair
// into LDS bitmaps, and every thread then walks its tile's bits in draw order, bumping `bump.segments` once per
// (tile, path) and `bump.ptcl` once per 256-word chunk.  On MI355X that shape is a few dozen workgroups (49 at
// 1600^2) of serial, one-wave-per-SIMD instruction streams and dependent global round trips on a 256-CU part, and the
// coverage pass reads the whole tile pool (a road map's long thin paths allocate 10 M tiles of which a third hold
// anything).  What the stage computes is kept (the commands of every tile, word for word and in draw order, the
// clip_zero_depth state machine, CMD_JUMP-linked storage); how is new:
//
//  * k_coarse_prep (one launch, two jobs by block range): (a) per draw object, everything coarse needs from five
//    buffers (tag, draw flags, first draw-data word, offsets, the Path record) gathered ONCE into a 32-byte record --
//    the reference re-gathers it per (bin, draw object); (b) per 64 tiles of the pool, three bits per tile by wave
//    ballot: has segments / backdrop == 0 / backdrop even -- all the coverage test needs (coarse.wgsl:318-341) -- stored
//    a word per 8 tiles with the three planes' bytes side by side (plane_window below).
//  * k_coarse: one workgroup of four waves per 8x8-tile QUADRANT of a bin (4x the workgroups; a draw object is only
//    considered by the quadrants its bbox touches).  Threads first act as draw objects: the bin's list is streamed
//    256 entries at a time (entries prefetched one round ahead), survivors are queued in LDS and each gets its 64-bit
//    coverage masks over the quadrant from <= 8 row windows of the bit planes (no Tile reads, no LDS atomics).
//  * PTCL storage is EXACT-FIT: per batch of 256 queued objects a tile gets one region of exactly the words it needs
//    (+ a two-word tail for the next CMD_JUMP / CMD_END, + slack for small follow-ups) instead of 256-word chunks grown
//    on demand.  That turns every offset into a prefix sum: command sizes follow from popcounts of bitmaps, so wave w
//    owns objects [64w, 64w + 64) of the batch and all four emit concurrently, one LANE PER (tile, object) PAIR --
//    Tile record loaded by the lane that needs it, segment slices from a segmented shuffle scan -- behind ONE atomic
//    per counter and batch.  The reference's per-tile walk survives only as a short sequential pre-pass over the clip
//    objects of scenes that have clips (the clip_zero_depth machine is order dependent; everything else is not).
//    fine follows CMD_JUMP wherever it points, so the command streams are the reference's, only the jump points move.
//  * occlusion culling (not in the reference): when a batch holds, for a tile, a fully covering OPAQUE solid-colour
//    draw (CMD_SOLID + CMD_COLOR with alpha 255) outside any clip/blend layer, premultiplied src-over makes everything
//    underneath irrelevant bit-exactly (x*0 + c == c), so the tile's list is restarted at that draw and the covered
//    draws of the batch are never emitted nor given segments.  Only for scenes without clips;
//    VELLO_HIP_DEBUG_NO_CULL turns it off for word-exact PTCL / segment diffs against the reference.
#include "engine.h"

namespace vk {

namespace {

constexpr uint32_t SUB_W = 8;         // a workgroup owns an 8x8-tile quadrant of a bin
#ifndef VK_COARSE_NW
#define VK_COARSE_NW 8  // (sweep constant: 4 = half the batch, half the queue -- 46 KB of LDS instead of 86)
#endif
constexpr uint32_t NW = VK_COARSE_NW; // waves per workgroup; wave w emits objects [64w, 64w + 64) of a batch
constexpr uint32_t WG = 64 * NW;
constexpr uint32_t NB = 64 * NW;      // draw objects per batch
constexpr uint32_t QCAP = 2 * NB;     // queue slots: NB - 1 left over + WG new ones; power of two
static_assert((QCAP & (QCAP - 1u)) == 0u && QCAP >= NB - 1u + WG, "the queue is a ring of a power of two that holds a batch's leftovers and a round's arrivals");
constexpr uint32_t PART_CHUNK = 256;  // bin headers (partitions of 256 draw objects) merged at a time
constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t EMIT_GROUP = 8;    // wave steps (64 pairs each) whose Tile loads are in flight together
constexpr uint32_t INITIAL_ROOM = PTCL_INITIAL_ALLOC - 1u - 2u - 1u;  // a tile's fixed block minus the blend word, the tail and the work word
constexpr uint32_t REGION_SLACK = 62u;  // words a new region holds beyond the batch that asked for it
// what a draw object emits per tile: a path command (CMD_FILL 4 words / CMD_SOLID 1) + a draw command of 2 or 3 words,
// or CMD_BEGIN_CLIP alone (coarse.wgsl:377-450)
constexpr uint32_t KIND_NONE = 0u, KIND_PATH2 = 1u, KIND_PATH3 = 2u, KIND_BEGIN = 3u;

typedef unsigned long long u64;

struct __attribute__((packed, aligned(4))) PtclWords2 { uint32_t a, b; };
struct __attribute__((packed, aligned(4))) PtclWords3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(4))) PtclWords4 { uint32_t a, b, c, d; };

// Queue of the draw objects that touch this quadrant, in draw order (a ring of QCAP slots), and the per-batch tables.
struct CoarseLds {
    uint32_t tag[QCAP];
    uint32_t flags[QCAP];   // draw_flags = info[di]
    uint32_t w0[QCAP];      // scene[dd]: colour / gradient index / blend mode
    uint32_t dd[QCAP];
    uint32_t di[QCAP];
    uint32_t base[QCAP];    // Tile index of quadrant-local (0, 0) in the path's tile rectangle (may lie outside it)
    uint32_t stride[QCAP];
    uint32_t kind[QCAP];    // KIND_*
    uint32_t cover[QCAP][8];  // 64-bit masks over the quadrant (bit = y * 8 + x): included, occluder, has segments, backdrop clear
    // per batch: [w][t] = wave-slice w (64 objects), tile t
    uint32_t em[NW][64][2];   // objects of the slice that emit commands for the tile
    uint32_t gm[NW][64][2];   // ... of those, the ones with segments (CMD_FILL)
    uint32_t cl[NW][64][2];   // backdrop-clear bits (clip pre-pass)
    uint32_t kslice[NW][64];  // 1 + slice-local index of the tile's last occluder, 0 = none
    uint32_t kinds[NW][2][2]; // per slice: which objects are BEGIN_CLIPs / END_CLIPs (clip pre-pass)
    uint32_t S[NW][64];             // words the slice needs in the tile
    uint32_t pend[NW][64];          // inclusive prefix over the tiles of the slice's pair counts
    uint32_t wordbase[NW][64];      // where the slice's commands for the tile start
    uint32_t part_end[PART_CHUNK];  // inclusive prefix of the element counts of the merged bin headers
    uint32_t part_off[PART_CHUNK];
    uint32_t wave_cnt[NW];
    uint32_t bcast;
    uint32_t seg_tot[NW], seg_base[NW], seg_arrive;  // a batch's first-group segment demand per slice, its answer, the waves that have reported
};

__device__ __forceinline__ uint32_t popc64(u64 x) { return (uint32_t)__popcll(x); }
__device__ __forceinline__ u64 make64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | (u64)lo; }
__device__ __forceinline__ u64 below64(uint32_t b) { return b >= 64u ? ~0ull : ((1ull << b) - 1ull); }

// position of the k-th (0-based) set bit of m; k < popcount(m)
__device__ __forceinline__ uint32_t kth_bit64(u64 m, uint32_t k) {
    uint32_t w = (uint32_t)m, pos = 0u;
    uint32_t c = (uint32_t)__popc(w);
    if (k >= c) { k -= c; w = (uint32_t)(m >> 32); pos = 32u; }
    c = (uint32_t)__popc(w & 0xffffu);
    if (k >= c) { k -= c; w >>= 16; pos += 16u; }
    c = (uint32_t)__popc(w & 0xffu);
    if (k >= c) { k -= c; w >>= 8; pos += 8u; }
    c = (uint32_t)__popc(w & 0xfu);
    if (k >= c) { k -= c; w >>= 4; pos += 4u; }
    c = (uint32_t)__popc(w & 3u);
    if (k >= c) { k -= c; w >>= 2; pos += 2u; }
    if (k >= (w & 1u)) pos += 1u;
    return pos;
}

// Transpose of a 64 x 64 bit matrix held one row per lane: afterwards bit e of lane t is what bit t of lane e was.
// Six butterfly steps (swap the off-diagonal j x j blocks between lanes r and r ^ j), 2 shuffles each.
__device__ __forceinline__ u64 transpose64(u64 x, uint32_t lane) {
    const u64 masks[6] = {0x00000000ffffffffull, 0x0000ffff0000ffffull, 0x00ff00ff00ff00ffull,
                          0x0f0f0f0f0f0f0f0full, 0x3333333333333333ull, 0x5555555555555555ull};
#pragma unroll
    for (uint32_t s = 0; s < 6u; s++) {
        const uint32_t j = 32u >> s;
        const u64 mask = masks[s];  // columns c with (c & j) == 0
        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)x, (int)j), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), (int)j);
        const u64 other = make64(olo, ohi);
        if ((lane & j) == 0u) x = (x & mask) | ((other & mask) << j);
        else x = (x & ~mask) | ((other & ~mask) >> j);
    }
    return x;
}

// words the objects of mask m (those of gmask with CMD_FILL) emit: FILL 4 / SOLID 1 + draw command 2 or 3; BEGIN 1
__device__ __forceinline__ uint32_t words_of(u64 m, u64 g, u64 k1, u64 k2, u64 k3) {
    return 3u * popc64(g) + 3u * popc64(m & k1) + 4u * popc64(m & k2) + popc64(m & k3);
}

// The three tile bits coarse.wgsl:318-341 needs (segments present / backdrop zero / backdrop even), a WORD per 8 tiles:
// byte 0 the eight tiles' "segments" bits, byte 1 "zero", byte 2 "even".  A quadrant is 8 tiles wide, so a row of an
// object's rectangle inside it is at most 8 consecutive tiles: words b / 8 and b / 8 + 1 hold all three planes' bits of
// the row -- ONE 8-byte load (4-byte aligned; two words of slack behind the last tile) where three separate planes took
// one per plane.  k_coarse's stream rounds are bound by the number of scattered requests a CU's address unit takes
// (512 threads x 8 rows x the planes read), not by their bytes.
__device__ __forceinline__ PtclWords2 plane_window(const uint32_t *__restrict__ bits, uint32_t b) {
    return *reinterpret_cast<const PtclWords2 *>(bits + (b >> 3));
}
// the 8-tile window of the plane whose bytes sit `byte_shift` bits up, starting at tile b (bits above the row: other tiles', masked by the caller)
__device__ __forceinline__ uint32_t window_bits(PtclWords2 w, uint32_t byte_shift, uint32_t b) {
    return (((w.a >> byte_shift) & 0xffu) | (((w.b >> byte_shift) & 0xffu) << 8)) >> (b & 7u);
}

}  // namespace

// Job (a), blocks [0, n_el_blocks): the per-draw-object record.  Job (b), the remaining blocks: the tile bit planes.
__global__ void __launch_bounds__(256) k_coarse_prep(Config cfg, uint32_t n_el_blocks, const uint32_t *__restrict__ scene,
                                                     const DrawMonoid *__restrict__ draw_monoids, const uint32_t *__restrict__ info_bin_data,
                                                     const Path *__restrict__ paths, const Tile *__restrict__ tiles, const Bump *__restrict__ bump,
                                                     CoarseEl *__restrict__ coarse_el, uint32_t *__restrict__ tile_bits) {
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x < n_el_blocks) {
        const uint32_t drawobj_ix = blockIdx.x * 256u + tid;
        if (drawobj_ix >= cfg.layout.n_draw_objects) return;
        CoarseEl e;
        e.tag = scene[cfg.layout.draw_tag_base + drawobj_ix];
        e.flags = 0u; e.w0 = 0u; e.dd = 0u; e.di = 0u; e.tiles = 0u; e.bbox_x = 0u; e.bbox_y = 0u;
        if (e.tag != DRAWTAG_NOP) {  // coarse.wgsl:264-289
            const DrawMonoid dm = draw_monoids[drawobj_ix];
            e.dd = cfg.layout.draw_data_base + dm.scene_offset;
            e.di = dm.info_offset;
            e.flags = info_bin_data[dm.info_offset];
            e.w0 = scene[e.dd];
            const Path path = paths[dm.path_ix];
            e.tiles = path.tiles;
            e.bbox_x = path.bbox[0] | (path.bbox[2] << 16);  // tile coordinates fit 16 bits (checked on the host)
            e.bbox_y = path.bbox[1] | (path.bbox[3] << 16);
        }
        coarse_el[drawobj_ix] = e;
        return;
    }
    // Tile -> three bits (coarse.wgsl:318-341 needs exactly these of a Tile): segments present / backdrop zero /
    // backdrop even.  64 tiles per wave step, coalesced 8-byte loads, ballots, aligned dword stores.
    const uint32_t n_tiles = minu(bump->tile, cfg.tiles_size);
    const uint32_t lane = tid & 63u;
    const uint32_t wave = (blockIdx.x - n_el_blocks) * 4u + (tid >> 6);
    const uint32_t n_waves = (gridDim.x - n_el_blocks) * 4u;
    for (uint32_t chunk = wave; chunk * 64u < n_tiles; chunk += n_waves) {
        const uint32_t i = chunk * 64u + lane;
        Tile t{1, 0u};
        if (i < n_tiles) t = tiles[i];
        const unsigned long long ms = __ballot(t.segment_count_or_ix != 0u);
        const unsigned long long mz = __ballot(t.backdrop == 0);
        const unsigned long long mo = __ballot((t.backdrop & 1) == 0);
        if (lane < 8u) {  // word j of the chunk: tiles [8j, 8j + 8) of it
            const uint32_t sh = 8u * lane;
            tile_bits[(size_t)chunk * 8u + lane] = ((uint32_t)(ms >> sh) & 0xffu) | (((uint32_t)(mz >> sh) & 0xffu) << 8) | (((uint32_t)(mo >> sh) & 0xffu) << 16);
        }
    }
}


// coarse.wgsl:156-471.  256 threads: as draw objects while the bin's list is filtered, as 4 x 64 (tile, object) pairs
// while a batch is emitted; wave 0's lanes are also the 64 tiles of the quadrant (write pointers, clip state).
__global__ void __launch_bounds__(WG) k_coarse(Config cfg, const uint32_t *__restrict__ scene, const BinHeader *__restrict__ bin_headers,
                                                const uint32_t *__restrict__ info_bin_data, const CoarseEl *__restrict__ coarse_el,
                                                const uint32_t *__restrict__ tile_bits, Tile *tiles, Bump *bump,
                                                uint32_t *ptcl, bool allow_cull, uint32_t *work_count, uint32_t *tile_order,
                                                SliceItem *slice_items, uint32_t *slice_counters, uint32_t slice_cap, uint32_t cov_cap,
                                                uint32_t slice_fills, uint32_t slice_min_fills) {
#ifdef VELLO_SIMT_EMU
    __shared__ CoarseLds sh;
#else
    // more than the 64 KB a kernel may declare statically: dynamic LDS, sized and enabled by launch_coarse
    extern __shared__ __attribute__((aligned(16))) unsigned char coarse_lds_raw[];
    CoarseLds &sh = *reinterpret_cast<CoarseLds *>(coarse_lds_raw);
#endif
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    {  // coarse.wgsl:161-176
        uint32_t failed = bump->failed & (STAGE_BINNING | STAGE_TILE_ALLOC | STAGE_FLATTEN | FAILED_SCENE);
        if (bump->seg_counts > cfg.seg_counts_size) failed |= STAGE_PATH_COUNT;
        if (failed != 0u) {
            if (blockIdx.x == 0 && tid == 0) atomicOr(&bump->failed, failed);
            return;
        }
    }
    const uint32_t width_in_bins = (cfg.width_in_tiles + N_TILE_X - 1u) / N_TILE_X;
    const uint32_t height_in_bins = (cfg.height_in_tiles + N_TILE_Y - 1u) / N_TILE_Y;
    const uint32_t n_bins = width_in_bins * height_in_bins;
    // workgroup -> (bin, quadrant): consecutive workgroup ids go to different XCDs (id mod 8), so the four quadrants
    // of a bin, which read the same bin lists and records, get ids that are equal mod 8 and share one L2
    const uint32_t bin_ix = ((blockIdx.x >> 3) >> 2) * 8u + (blockIdx.x & 7u);
    const uint32_t quad = (blockIdx.x >> 3) & 3u;
    if (bin_ix >= n_bins) return;
    const uint32_t aligned_n_bins = (n_bins + N_TILE - 1u) & ~(N_TILE - 1u);
    const uint32_t n_partitions = (cfg.layout.n_draw_objects + N_TILE - 1u) / N_TILE;
    const uint32_t sub_x0 = N_TILE_X * (bin_ix % width_in_bins) + SUB_W * (quad & 1u);
    const uint32_t sub_y0 = N_TILE_Y * (bin_ix / width_in_bins) + SUB_W * (quad >> 1);
    const bool has_clips = cfg.layout.n_clips != 0u;
    const bool cull = allow_cull && !has_clips;
    const uint32_t ptcl_dyn_start = cfg.width_in_tiles * cfg.height_in_tiles * PTCL_INITIAL_ALLOC;
    if (tid == 0u) sh.seg_arrive = 0u;  // (read behind the barriers of the first batch)
#ifdef VELLO_COARSE_PROF
    // Measurement build only (scripts/coarse_prof.py): shader-clock cycles per phase and workgroup, left in the last
    // 8192 words of the PTCL pool.  0 stream rounds, 1 batch front (transposes, occluders, clips), 2 allocation,
    // 3 emission, 4 batch end, 5 list end; 6 = stream rounds, 7 = batches.
    long long prof_prev = clock64();
    uint32_t prof_acc[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#define CPROF(k) do { const long long t_ = clock64(); prof_acc[k] += (uint32_t)(t_ - prof_prev); prof_prev = t_; } while (0)
#define CPROF_COUNT(k) prof_acc[k] += 1u
#else
#define CPROF(k)
#define CPROF_COUNT(k)
#endif

    // ---- per-tile state, live in wave 0 (lane = tile) ----
    const uint32_t this_tile_ix = (sub_y0 + lane / SUB_W) * cfg.width_in_tiles + sub_x0 + lane % SUB_W;
    const uint32_t blend_offset = this_tile_ix * PTCL_INITIAL_ALLOC;
    const uint32_t list_start = blend_offset + 1u;
    uint32_t cur = list_start;       // next command word of the tile's list
    uint32_t room = INITIAL_ROOM;    // words left in the current region in front of its two-word tail
    bool dead = false;               // the tile's list ran out of PTCL pool: nothing more is written
    uint32_t words_total = 0u;       // command words of the tile's list so far
    uint32_t fills_total = 0u;       // CMD_FILLs among them
    uint32_t clip_zero_depth = 0u, clip_depth = 0u, render_blend_depth = 0u, max_blend_depth = 0u;

    // ---- the bin's element stream: bin headers merged PART_CHUNK partitions at a time (coarse.wgsl:218-263) ----
    uint32_t partition_ix = 0u;                 // next partition to merge
    uint32_t chunk_total = 0u, chunk_pos = 0u;  // elements of the merged partitions / handed out so far
    auto refill = [&]() {
        __syncthreads();  // the binary searches of the previous round are done with part_end / part_off
        if (wave == 0u) {
            uint32_t carry = 0u;
#pragma unroll
            for (uint32_t k = 0; k < PART_CHUNK / 64u; k++) {
                const uint32_t p = partition_ix + k * 64u + lane;
                uint32_t count = 0u, off = 0u;
                if (p < n_partitions) {
                    const BinHeader bh = bin_headers[(size_t)p * aligned_n_bins + bin_ix];
                    count = bh.element_count;
                    off = bh.chunk_offset;
                }
                const uint32_t incl = wave_incl_scan_u32(count, (int)lane) + carry;
                sh.part_end[k * 64u + lane] = incl;
                sh.part_off[k * 64u + lane] = off;
                carry = (uint32_t)__shfl((int)incl, 63);
            }
            if (lane == 0u) sh.bcast = carry;
        }
        __syncthreads();
        chunk_total = sh.bcast;
        chunk_pos = 0u;
        partition_ix += PART_CHUNK;
    };
    // the draw object index of this thread's entry of the next round (NONE = none); false when the stream has ended
    auto fetch_index = [&](uint32_t &d) -> bool {
        while (chunk_pos >= chunk_total && partition_ix < n_partitions) refill();
        const bool any = chunk_pos < chunk_total;
        const uint32_t ix = chunk_pos + tid;
        d = NONE;
        if (ix < chunk_total) {
            uint32_t part = 0u;
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) {
                const uint32_t probe = part + (128u >> i);
                if (ix >= sh.part_end[probe - 1u]) part = probe;
            }
            const uint32_t local = ix - (part > 0u ? sh.part_end[part - 1u] : 0u);
            d = info_bin_data[cfg.layout.bin_data_start + sh.part_off[part] + local];
        }
        chunk_pos = minu(chunk_pos + WG, chunk_total);
        return any;
    };

    uint32_t qh = 0u, qlen = 0u;  // queue head (ring index) and length
    // The stream runs two rounds ahead of its consumer: while round r is filtered and gets its coverage masks (one
    // memory round trip for the bit-plane windows), the records of round r + 1 and the list entries of round r + 2 are
    // already in flight.
    auto load_el = [&](uint32_t d) -> CoarseEl {
        CoarseEl e;
        e.tag = DRAWTAG_NOP;
        e.flags = 0u; e.w0 = 0u; e.dd = 0u; e.di = 0u; e.tiles = 0u; e.bbox_x = 0u; e.bbox_y = 0u;
        if (d != NONE) {
            const uint4 *p = reinterpret_cast<const uint4 *>(coarse_el + d);
            const uint4 a = p[0], b = p[1];
            e.tag = a.x; e.flags = a.y; e.w0 = a.z; e.dd = a.w;
            e.di = b.x; e.tiles = b.y; e.bbox_x = b.z; e.bbox_y = b.w;
        }
        return e;
    };
    uint32_t d_next;
    bool cur_valid = fetch_index(d_next);  // round 0
    CoarseEl el_next = load_el(d_next);
    bool next_valid = cur_valid ? fetch_index(d_next) : false;  // entries of round 1
    bool more = cur_valid;
    while (cur_valid || qlen > 0u) {
        if (cur_valid) {
            // ---- threads as draw objects ----
            const CoarseEl el = el_next;
            if (next_valid) el_next = load_el(d_next);                  // records of the next round
            const bool nn_valid = next_valid ? fetch_index(d_next) : false;  // entries of the round after it
            cur_valid = next_valid;
            next_valid = nn_valid;
            more = cur_valid;
            CPROF(8);
            // An object matters to this quadrant if its tile rectangle meets it (coarse.wgsl:264-289) AND at least one of
            // those tiles is included (coarse.wgsl:318-341): its 64-bit coverage masks come from <= 8 row windows of the
            // bit planes (the reference reads a Tile per (object, tile) pair and sets LDS bits with atomics), and an
            // object whose mask is empty -- most of a road's bounding box -- is never queued.
            const int32_t bx0 = (int32_t)(el.bbox_x & 0xffffu), bx1 = (int32_t)(el.bbox_x >> 16);
            const int32_t by0 = (int32_t)(el.bbox_y & 0xffffu), by1 = (int32_t)(el.bbox_y >> 16);
            const int32_t dx = bx0 - (int32_t)sub_x0, dy = by0 - (int32_t)sub_y0;
            const int32_t x0 = clampi(dx, 0, (int32_t)SUB_W), y0 = clampi(dy, 0, (int32_t)SUB_W);
            const int32_t x1 = clampi(bx1 - (int32_t)sub_x0, 0, (int32_t)SUB_W), y1 = clampi(by1 - (int32_t)sub_y0, 0, (int32_t)SUB_W);
            const bool meets = el.tag != DRAWTAG_NOP && x1 > x0 && y1 > y0;
            const uint32_t stride = (uint32_t)(bx1 - bx0);
            const uint32_t base = el.tiles - (uint32_t)(dy * (int32_t)stride + dx);
            u64 inc = 0ull, kil = 0ull, seg = 0ull, clr = 0ull;
            {
                // BRANCH-FREE on purpose: with each row's load under `if (r < h)` the compiler ends every one of the
                // conditional blocks with s_waitcnt vmcnt(0) -- a memory round trip per row, one after the other (round 2,
                // when a row was two loads: 8 300 of a round's 13 000 cycles).  Rows an object does not have (and objects that miss the quadrant) load the
                // first words of the plane instead and get an empty row mask.
                const uint32_t rx0 = (uint32_t)x0, ry0 = (uint32_t)y0;
                const uint32_t w = meets ? (uint32_t)(x1 - x0) : 0u, h = meets ? (uint32_t)(y1 - y0) : 0u;
                const bool is_clip = (el.tag & 1u) != 0u;
                const uint32_t BLEND_CLIP = (128u << 8) | 3u;
                const bool is_blend = is_clip && el.w0 != BLEND_CLIP;
                const bool even_odd = (el.flags & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u;
                const uint32_t clear_byte = even_odd ? 16u : 8u;  // which plane says "backdrop clear" under the object's fill rule
                const uint32_t wmask = (1u << w) - 1u;
                const uint32_t b0 = base + stride * ry0 + rx0;
                PtclWords2 win[SUB_W];
#pragma unroll
                for (uint32_t r = 0; r < SUB_W; r++) win[r] = plane_window(tile_bits, r < h ? b0 + stride * r : 0u);
#pragma unroll
                for (uint32_t r = 0; r < SUB_W; r++) {
                    const uint32_t rmask = r < h ? wmask : 0u;
                    const uint32_t b = b0 + stride * r;  // (only its low three bits are used: rows the object does not have are masked)
                    const uint32_t sg = window_bits(win[r], 0u, b) & rmask;
                    const uint32_t clear = window_bits(win[r], clear_byte, b) & rmask;  // backdrop_clear per tile of the row
                    // include_tile = n_segs != 0 || (backdrop_clear == is_clip) || is_blend
                    const uint32_t in = is_blend ? rmask : (sg | ((is_clip ? clear : ~clear) & rmask));
                    const uint32_t shift = ((ry0 + r) * SUB_W + rx0) & 63u;
                    inc |= (u64)in << shift;
                    seg |= (u64)(in & sg) << shift;
                    clr |= (u64)(in & clear) << shift;
                }
                // fully covering opaque solid colour: occludes every earlier draw of the tile
                if (cull && el.tag == DRAWTAG_FILL_COLOR && (el.w0 >> 24) == 0xffu) kil = inc & ~seg;
            }
            const bool keep = inc != 0ull;
            const u64 m = __ballot(keep);
            CPROF(9);
            if (lane == 0u) sh.wave_cnt[wave] = popc64(m);
            __syncthreads();
            CPROF(10);
            uint32_t before = 0u, total_new = 0u;
#pragma unroll
            for (uint32_t w = 0; w < NW; w++) {
                const uint32_t c = sh.wave_cnt[w];
                if (w < wave) before += c;
                total_new += c;
            }
            if (keep) {
                const uint32_t q = (qh + qlen + before + popc64(m & below64(lane))) & (QCAP - 1u);
                uint32_t kind = KIND_NONE;
                switch (el.tag) {
                case DRAWTAG_FILL_COLOR: case DRAWTAG_FILL_IMAGE: kind = KIND_PATH2; break;
                case DRAWTAG_BLURRED_ROUNDED_RECT: case DRAWTAG_FILL_LIN_GRADIENT: case DRAWTAG_FILL_RAD_GRADIENT:
                case DRAWTAG_FILL_SWEEP_GRADIENT: case DRAWTAG_END_CLIP: kind = KIND_PATH3; break;
                case DRAWTAG_BEGIN_CLIP: kind = KIND_BEGIN; break;
                default: break;
                }
                sh.tag[q] = el.tag;
                sh.flags[q] = el.flags;
                sh.w0[q] = el.w0;
                sh.dd[q] = el.dd;
                sh.di[q] = el.di;
                sh.stride[q] = stride;
                sh.base[q] = base;
                sh.kind[q] = kind;
                sh.cover[q][0] = (uint32_t)inc; sh.cover[q][1] = (uint32_t)(inc >> 32);
                sh.cover[q][2] = (uint32_t)kil; sh.cover[q][3] = (uint32_t)(kil >> 32);
                sh.cover[q][4] = (uint32_t)seg; sh.cover[q][5] = (uint32_t)(seg >> 32);
                sh.cover[q][6] = (uint32_t)clr; sh.cover[q][7] = (uint32_t)(clr >> 32);
            }
            qlen += total_new;
            __syncthreads();
            CPROF(0);
            CPROF_COUNT(6);
        }

        // ---- full batches, and whatever is left when the stream has ended ----
        while (qlen >= NB || (!more && qlen > 0u)) {
            const uint32_t n = minu(qlen, NB);
            const uint32_t cnt = n > wave * 64u ? minu(n - wave * 64u, 64u) : 0u;  // objects of this wave's slice
            const uint32_t q0 = qh + wave * 64u;
            // lanes as objects of the slice: which kinds are where (wave-uniform masks)
            const uint32_t my_kind = lane < cnt ? (sh.kind[(q0 + lane) & (QCAP - 1u)]) : KIND_NONE;
            const u64 k1 = __ballot(my_kind == KIND_PATH2), k2 = __ballot(my_kind == KIND_PATH3), k3 = __ballot(my_kind == KIND_BEGIN);
            const u64 k_end = has_clips ? __ballot(lane < cnt && sh.tag[(q0 + lane) & (QCAP - 1u)] == DRAWTAG_END_CLIP) : 0ull;
            // lanes as objects -> lanes as tiles: the slice's 64 coverage masks (one per object, a bit per tile) become
            // 64 bitmaps (one per tile, a bit per object) by a 64 x 64 bit-matrix transpose in registers
            u64 inc = 0ull, seg = 0ull, clr = 0ull, kil = 0ull;
            if (lane < cnt) {
                const uint32_t *c = sh.cover[(q0 + lane) & (QCAP - 1u)];
                inc = make64(c[0], c[1]);
                if (cull) kil = make64(c[2], c[3]);
                seg = make64(c[4], c[5]);
                if (has_clips) clr = make64(c[6], c[7]);
            }
            inc = transpose64(inc, lane);
            seg = transpose64(seg, lane);
            if (cull) kil = transpose64(kil, lane);
            if (has_clips) clr = transpose64(clr, lane);
            const uint32_t klast = kil != 0ull ? 64u - (uint32_t)__clzll((long long)kil) : 0u;  // 1 + the tile's last occluder
            sh.kslice[wave][lane] = klast;
            if (has_clips) {
                sh.em[wave][lane][0] = (uint32_t)inc; sh.em[wave][lane][1] = (uint32_t)(inc >> 32);
                sh.gm[wave][lane][0] = (uint32_t)seg; sh.gm[wave][lane][1] = (uint32_t)(seg >> 32);
                sh.cl[wave][lane][0] = (uint32_t)clr; sh.cl[wave][lane][1] = (uint32_t)(clr >> 32);
                if (lane == 0u) {
                    sh.kinds[wave][0][0] = (uint32_t)k3; sh.kinds[wave][0][1] = (uint32_t)(k3 >> 32);
                    sh.kinds[wave][1][0] = (uint32_t)k_end; sh.kinds[wave][1][1] = (uint32_t)(k_end >> 32);
                }
            }
            __syncthreads();  // (1) occluders / clip inputs of all slices visible
            bool tile_killed = false;  // the tile's list restarts in this batch
            if (cull) {
                // the last occluder of the tile in the batch: earlier objects are dropped
                bool later = false;
#pragma unroll
                for (uint32_t w = 0; w < NW; w++) {
                    const uint32_t kl = sh.kslice[w][lane];
                    tile_killed = tile_killed || kl != 0u;
                    if (w > wave && kl != 0u) later = true;
                }
                if (later) inc = 0ull;
                else if (klast != 0u) inc &= ~below64(klast - 1u);
            }
            if (has_clips) {
                // The clip state machine is order dependent (coarse.wgsl:416-450): wave 0 walks, per tile, the CLIP objects
                // of the whole batch in order and clears the bits of everything that does not emit.
                if (wave == 0u) {
#pragma unroll 1
                    for (uint32_t w = 0; w < NW; w++) {
                        const u64 m = make64(sh.em[w][lane][0], sh.em[w][lane][1]);
                        const u64 sg = make64(sh.gm[w][lane][0], sh.gm[w][lane][1]);
                        const u64 cr = make64(sh.cl[w][lane][0], sh.cl[w][lane][1]);
                        const u64 kb = make64(sh.kinds[w][0][0], sh.kinds[w][0][1]);  // BEGIN_CLIPs of the slice
                        const u64 ke = make64(sh.kinds[w][1][0], sh.kinds[w][1][1]);  // END_CLIPs
                        u64 clipbits = m & (kb | ke);
                        u64 act = 0ull;
                        uint32_t pos = 0u;
                        while (clipbits != 0ull) {
                            const uint32_t b = (uint32_t)__ffsll((long long)clipbits) - 1u;
                            clipbits &= clipbits - 1ull;
                            if (clip_zero_depth == 0u) act |= below64(b) & ~below64(pos);
                            const bool is_begin = ((kb >> b) & 1ull) != 0ull;
                            if (clip_zero_depth == 0u) {
                                if (is_begin) {
                                    const bool empty = ((sg >> b) & 1ull) == 0ull && ((cr >> b) & 1ull) != 0ull;
                                    if (empty) {
                                        clip_zero_depth = clip_depth + 1u;
                                    } else {
                                        act |= 1ull << b;
                                        render_blend_depth += 1u;
                                        max_blend_depth = maxu(max_blend_depth, render_blend_depth);
                                    }
                                    clip_depth += 1u;
                                } else {
                                    clip_depth -= 1u;
                                    act |= 1ull << b;
                                    render_blend_depth -= 1u;
                                }
                            } else {
                                if (is_begin) {
                                    clip_depth += 1u;
                                } else {
                                    if (clip_depth == clip_zero_depth) clip_zero_depth = 0u;
                                    clip_depth -= 1u;
                                }
                            }
                            pos = b + 1u;
                        }
                        if (clip_zero_depth == 0u) act |= ~below64(pos);
                        const u64 e = m & act;
                        sh.em[w][lane][0] = (uint32_t)e; sh.em[w][lane][1] = (uint32_t)(e >> 32);
                    }
                }
                __syncthreads();
                inc = make64(sh.em[wave][lane][0], sh.em[wave][lane][1]);
            }
            // what the slice emits for this tile
            const u64 em = inc & (k1 | k2 | k3);
            const u64 gmask = seg & em & (k1 | k2);
            const uint32_t n_pairs_t = popc64(em);
            const uint32_t pair_incl = wave_incl_scan_u32(n_pairs_t, (int)lane);
            const uint32_t total_pairs = (uint32_t)__shfl((int)pair_incl, 63);
            sh.em[wave][lane][0] = (uint32_t)em; sh.em[wave][lane][1] = (uint32_t)(em >> 32);
            sh.gm[wave][lane][0] = (uint32_t)gmask; sh.gm[wave][lane][1] = (uint32_t)(gmask >> 32);
            sh.pend[wave][lane] = pair_incl;
            sh.S[wave][lane] = words_of(em, gmask, k1, k2, k3) | (popc64(gmask) << 16);  // (<= 448 words, <= 64 fills)
            wave_lds_sync();
            const uint32_t n_iter = (total_pairs + 63u) / 64u;
            // pair p of the slice, tile-major: which tile, which object, does it have segments
            auto pair_of = [&](uint32_t p, uint32_t &t, uint32_t &b, bool &has_segs) {
                t = 0u;
#pragma unroll
                for (uint32_t i = 0; i < 6u; i++) {
                    const uint32_t probe = t + (32u >> i);
                    if (p >= sh.pend[wave][probe - 1u]) t = probe;
                }
                const uint32_t k = p - (t > 0u ? sh.pend[wave][t - 1u] : 0u);
                const u64 m = make64(sh.em[wave][t][0], sh.em[wave][t][1]);
                b = kth_bit64(m, k);
                has_segs = ((make64(sh.gm[wave][t][0], sh.gm[wave][t][1]) >> b) & 1ull) != 0ull;
            };
            // the Tile records of a group's pairs, all requested before the first is used
            auto load_group = [&](uint32_t g0, uint32_t (&info)[EMIT_GROUP], uint32_t (&tix)[EMIT_GROUP], Tile (&tl)[EMIT_GROUP]) {
#pragma unroll
                for (uint32_t u = 0; u < EMIT_GROUP; u++) {
                    const uint32_t p = (g0 + u) * 64u + lane;
                    info[u] = 0u;
                    tix[u] = 0u;
                    tl[u] = Tile{0, 0u};
                    if (g0 + u < n_iter && p < total_pairs) {
                        uint32_t t, b;
                        bool has_segs;
                        pair_of(p, t, b, has_segs);
                        const uint32_t q = (q0 + b) & (QCAP - 1u);
                        tix[u] = sh.base[q] + sh.stride[q] * (t / SUB_W) + (t % SUB_W);
                        info[u] = t | (b << 8) | (has_segs ? 1u << 16 : 0u) | (1u << 17);
                        if (has_segs) tl[u] = tiles[tix[u]];
                    }
                }
            };
            uint32_t info[EMIT_GROUP], tix[EMIT_GROUP];  // t | b << 8 | has_segs << 16 | valid << 17; Tile index
            Tile tl[EMIT_GROUP];
            load_group(0u, info, tix, tl);  // in flight across the barrier and wave 0's allocation
            __syncthreads();  // (2) S of all slices
            CPROF(1);
            // ALLOCATE (wave 0, lane = tile): one PTCL region per tile that needs one behind ONE atomic; the command sizes
            // follow from the bitmaps alone, so no Tile has been read yet
            if (wave == 0u) {
                uint32_t s_w[NW], W = 0u, n_fill = 0u;
#pragma unroll
                for (uint32_t w = 0; w < NW; w++) {
                    const uint32_t s = sh.S[w][lane];
                    s_w[w] = s & 0xffffu;
                    W += s_w[w];
                    n_fill += s >> 16;
                }
                if (tile_killed && !dead) {  // everything emitted so far is covered: restart the list in the tile's own block
                    cur = list_start;
                    room = INITIAL_ROOM;
                    words_total = 0u;
                    fills_total = 0u;
                }
                words_total += W;
                fills_total += n_fill;
                const bool need = W > room && !dead;
                const uint32_t rsize = need ? W + 2u + REGION_SLACK : 0u;
                const uint32_t r_incl = wave_incl_scan_u32(rsize, (int)lane);
                const uint32_t r_total = (uint32_t)__shfl((int)r_incl, 63);
                uint32_t r_base = 0u;
                if (lane == 0u && r_total != 0u) r_base = atomicAdd(&bump->ptcl, r_total);
                r_base = (uint32_t)__shfl((int)r_base, 0);
                if (need) {
                    const uint32_t start = ptcl_dyn_start + r_base + (r_incl - rsize);
                    if (start < ptcl_dyn_start || start + rsize > cfg.ptcl_size || start + rsize < start) {
                        atomicOr(&bump->failed, STAGE_COARSE);
                        dead = true;
                    } else {
                        *reinterpret_cast<PtclWords2 *>(ptcl + cur) = PtclWords2{CMD_JUMP, start};
                        cur = start;
                        room = W + REGION_SLACK;
                    }
                }
                uint32_t wb = dead ? NONE : cur;
#pragma unroll
                for (uint32_t w = 0; w < NW; w++) {
                    sh.wordbase[w][lane] = wb;
                    if (!dead) wb += s_w[w];
                }
                if (!dead) {
                    cur += W;
                    room -= W;
                }
            }
            // EMIT: one lane per (tile, object) pair, EMIT_GROUP wave steps at a time: the segment counts of the group's Tile
            // records are scanned once and the wave reserves the
            // group's segment slices with ONE atomic (a slice may sit anywhere: CMD_FILL carries its index)
            // The first group's slices of ALL eight waves behind one atomic: bump.segments is one address, an atomic on it is a
            // queue of every wave of the launch that wants one (12 ns each), and a batch is a burst of them -- 1 400 on the road
            // map, the last of which waited 16 us.  Every wave leaves its demand in LDS; the one that reports last adds them
            // up and asks.
            CPROF(2);
            uint32_t seg_first;
            {
                uint32_t my_segs = 0u;
#pragma unroll
                for (uint32_t u = 0; u < EMIT_GROUP; u++) my_segs += tl[u].segment_count_or_ix;  // (no pairs: zeros)
                const uint32_t seg_incl = wave_incl_scan_u32(my_segs, (int)lane);
                const uint32_t seg_total = (uint32_t)__shfl((int)seg_incl, 63);
                uint32_t arrived = 0u;
                if (lane == 0u) {
                    sh.seg_tot[wave] = seg_total;
#ifdef VELLO_SIMT_EMU
                    arrived = atomicAdd(&sh.seg_arrive, 1u);
#else
                    // (release: the demand above is in LDS before the count says so; acquire: the last wave reads the others' behind it)
                    arrived = __hip_atomic_fetch_add(&sh.seg_arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                }
                arrived = (uint32_t)__shfl((int)arrived, 0);
                if (arrived == NW - 1u) {
                    const uint32_t tot = lane < NW ? sh.seg_tot[lane] : 0u;
                    const uint32_t incl = wave_incl_scan_u32(tot, (int)lane);
                    const uint32_t all = (uint32_t)__shfl((int)incl, 63);
                    uint32_t got = 0u;
                    if (lane == 0u && all != 0u) got = atomicAdd(&bump->segments, all);
                    got = (uint32_t)__shfl((int)got, 0);
                    if (lane < NW) sh.seg_base[lane] = got + (incl - tot);
                    if (lane == 0u) sh.seg_arrive = 0u;
                }
                __syncthreads();  // (3) wave 0 has published the word bases, the last wave the segment bases
                seg_first = sh.seg_base[wave] + (seg_incl - my_segs);
            }
            for (uint32_t g0 = 0; g0 < n_iter; g0 += EMIT_GROUP) {
                uint32_t seg_next = seg_first;
                if (g0 != 0u) {  // (more than 512 pairs in a slice: seldom; a reservation of the wave's own)
                    load_group(g0, info, tix, tl);
                    uint32_t my_segs = 0u;
#pragma unroll
                    for (uint32_t u = 0; u < EMIT_GROUP; u++) my_segs += tl[u].segment_count_or_ix;
                    const uint32_t seg_incl = wave_incl_scan_u32(my_segs, (int)lane);
                    const uint32_t seg_total = (uint32_t)__shfl((int)seg_incl, 63);
                    uint32_t got = 0u;
                    if (lane == 0u && seg_total != 0u) got = atomicAdd(&bump->segments, seg_total);
                    seg_next = (uint32_t)__shfl((int)got, 0) + (seg_incl - my_segs);
                }
#pragma unroll
                for (uint32_t u = 0; u < EMIT_GROUP; u++) {
                    if ((info[u] >> 17) == 0u) continue;
                    const uint32_t t = info[u] & 0xffu, b = (info[u] >> 8) & 0xffu;
                    const bool has_segs = ((info[u] >> 16) & 1u) != 0u;
                    const uint32_t n_segs = tl[u].segment_count_or_ix;
                    const uint32_t seg_ix = seg_next;
                    seg_next += n_segs;
                    const uint32_t wbase = sh.wordbase[wave][t];
                    if (wbase == NONE) continue;
                    const uint32_t q = (q0 + b) & (QCAP - 1u);
                    const u64 m = make64(sh.em[wave][t][0], sh.em[wave][t][1]);
                    const u64 g = make64(sh.gm[wave][t][0], sh.gm[wave][t][1]);
                    const u64 bel = below64(b);
                    uint32_t off = wbase + words_of(m & bel, g & bel, k1, k2, k3);
                    const uint32_t tag = sh.tag[q];
                    if (tag == DRAWTAG_BEGIN_CLIP) {
                        ptcl[off] = CMD_BEGIN_CLIP;
                        continue;
                    }
                    const uint32_t draw_flags = sh.flags[q];
                    if (has_segs) {  // coarse.wgsl:88-110
                        tiles[tix[u]].segment_count_or_ix = ~seg_ix;
                        const uint32_t even_odd = (draw_flags & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u ? 1u : 0u;
                        *reinterpret_cast<PtclWords4 *>(ptcl + off) = PtclWords4{CMD_FILL, (n_segs << 1) | even_odd, seg_ix, (uint32_t)tl[u].backdrop};
                        off += 4u;
                    } else {
                        ptcl[off] = CMD_SOLID;
                        off += 1u;
                    }
                    const uint32_t w0 = sh.w0[q], di = sh.di[q];
                    switch (tag) {  // coarse.wgsl:377-450
                    case DRAWTAG_FILL_COLOR: *reinterpret_cast<PtclWords2 *>(ptcl + off) = PtclWords2{CMD_COLOR, w0}; break;
                    case DRAWTAG_FILL_IMAGE: *reinterpret_cast<PtclWords2 *>(ptcl + off) = PtclWords2{CMD_IMAGE, di + 1u}; break;
                    case DRAWTAG_BLURRED_ROUNDED_RECT: *reinterpret_cast<PtclWords3 *>(ptcl + off) = PtclWords3{CMD_BLUR_RECT, di + 1u, w0}; break;
                    case DRAWTAG_FILL_LIN_GRADIENT: *reinterpret_cast<PtclWords3 *>(ptcl + off) = PtclWords3{CMD_LIN_GRAD, w0, di + 1u}; break;
                    case DRAWTAG_FILL_RAD_GRADIENT: *reinterpret_cast<PtclWords3 *>(ptcl + off) = PtclWords3{CMD_RAD_GRAD, w0, di + 1u}; break;
                    case DRAWTAG_FILL_SWEEP_GRADIENT: *reinterpret_cast<PtclWords3 *>(ptcl + off) = PtclWords3{CMD_SWEEP_GRAD, w0, di + 1u}; break;
                    case DRAWTAG_END_CLIP: *reinterpret_cast<PtclWords3 *>(ptcl + off) = PtclWords3{CMD_END_CLIP, w0, scene[sh.dd[q] + 1u]}; break;
                    default: break;
                    }
                }
            }
            CPROF(3);
            qh = (qh + n) & (QCAP - 1u);
            qlen -= n;
            __syncthreads();  // (4) the per-batch tables and the freed queue slots are rewritten next
            CPROF(4);
            CPROF_COUNT(7);
        }
    }
    if (wave == 0u) {
        const bool in_target = sub_x0 + lane % SUB_W < cfg.width_in_tiles && sub_y0 + lane / SUB_W < cfg.height_in_tiles;
        if (in_target) {
            if (!dead) ptcl[cur] = CMD_END;
            uint32_t blend_ix = 0u;
            if (max_blend_depth > BLEND_STACK_SPLIT) {
                uint32_t scratch_size = (max_blend_depth - BLEND_STACK_SPLIT) * TILE_WIDTH * TILE_HEIGHT;
                blend_ix = atomicAdd(&bump->blend, scratch_size);
                if (blend_ix + scratch_size > cfg.blend_size) atomicOr(&bump->failed, STAGE_COARSE);
            }
            ptcl[blend_offset] = blend_ix;
            // For k_fine: the length of the list in the last word of the tile's fixed block (it arrives with fine's
            // first window) ...
            ptcl[blend_offset + PTCL_INITIAL_ALLOC - 1u] = words_total;
        }
        // ... and the tile's index in the bucket of its length class, so that long lists are started first (one atomic
        // per bucket and workgroup: 10 000 tiles bumping eight counters one by one cost the kernel 50 us)
        const uint32_t n_tiles = cfg.width_in_tiles * cfg.height_in_tiles;
        uint32_t bucket = in_target ? minu(FINE_WORK_BUCKETS - 1u, words_total >> 5) : NONE;
        // A long list (MSAA modes) is cut into slices of slice_fills FILLs, each a work item of fine's that comes before all
        // unsliced tiles; the tile then stays out of the buckets.  One atomic per counter and workgroup; a tile whose slices or
        // coverage scratch do not fit is rendered unsliced, its place in the item array marked as a hole.
        uint32_t n_sl = 0u;
        if (in_target && !dead && slice_min_fills != 0u && fills_total >= slice_min_fills) n_sl = minu((fills_total + slice_fills - 1u) / slice_fills, 0xffffu);
        if (__ballot(n_sl != 0u) != 0ull) {
            const uint32_t cov_need = n_sl != 0u ? n_sl * slice_fills * 64u + 64u : 0u;  // 64 words per FILL + one fill of slack
            const uint32_t it_incl = wave_incl_scan_u32(n_sl, (int)lane), cv_incl = wave_incl_scan_u32(cov_need, (int)lane);
            uint32_t it_base = 0u, cv_base = 0u;
            if (lane == 63u) {
                it_base = atomicAdd(&work_count[FINE_WORK_BUCKETS], it_incl);      // Control::slice_items
                cv_base = atomicAdd(&work_count[FINE_WORK_BUCKETS + 1u], cv_incl);  // Control::cov_words
            }
            it_base = (uint32_t)__shfl((int)it_base, 63) + (it_incl - n_sl);
            cv_base = (uint32_t)__shfl((int)cv_base, 63) + (cv_incl - cov_need);
            if (n_sl != 0u) {
                const bool fits = it_base + n_sl <= slice_cap && it_base + n_sl >= it_base && cv_base + cov_need <= cov_cap && cv_base + cov_need >= cv_base;
                for (uint32_t k = 0; k < n_sl && it_base + k < slice_cap && it_base + k >= it_base; k++)
                    slice_items[it_base + k] = SliceItem{fits ? this_tile_ix : ~0u, k | (n_sl << 16), cv_base, it_base};
                if (fits) {
                    slice_counters[it_base] = 0u;
                    bucket = NONE;
                }
            }
        }
        // (lane b owns bucket b: all the buckets' atomics are ONE instruction and one round trip, not one after the other)
        static_assert(FINE_WORK_BUCKETS <= 64u, "a lane per bucket");
        uint32_t my_count = 0u, my_rank = 0u;
        for (uint32_t bk = 0; bk < FINE_WORK_BUCKETS; bk++) {
            const u64 m = __ballot(bucket == bk);
            if (lane == bk) my_count = popc64(m);
            if (bucket == bk) my_rank = popc64(m & below64(lane));
        }
        uint32_t base = 0u;
        if (lane < FINE_WORK_BUCKETS && my_count != 0u) base = atomicAdd(&work_count[lane], my_count);
        base = (uint32_t)__shfl((int)base, (int)(bucket & 63u));
        if (bucket != NONE) {
            const uint32_t pos = base + my_rank;
            if (pos < n_tiles) tile_order[bucket * n_tiles + pos] = this_tile_ix;
        }
    }
#ifdef VELLO_COARSE_PROF
    CPROF(5);
    if (tid == 0u && cfg.ptcl_size >= 8192u && blockIdx.x < 512u)
        for (uint32_t k = 0; k < 16u; k++) ptcl[cfg.ptcl_size - 8192u + blockIdx.x * 16u + k] = prof_acc[k];
#endif
}

// k_coarse's dynamic LDS is beyond the 64 KB a kernel gets without asking; function attributes are per device, so every
// context asks once for its own (vello_hip_create, after hipSetDevice) and reports a refusal there.
int enable_coarse_lds() {
#ifndef VELLO_SIMT_EMU
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(k_coarse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CoarseLds));
#else
    return 0;
#endif
}

void launch_coarse(const Frame &f, hipStream_t s, hipEvent_t *mid) {
    const uint32_t wb = (f.cfg.width_in_tiles + 15u) / 16u, hb = (f.cfg.height_in_tiles + 15u) / 16u;
    if (wb * hb == 0) {
        if (mid) (void)hipEventRecord(mid[0], s);  // (recorded on every way out: vello_hip_get_kernel_ms reads it)
        return;
    }
    const uint32_t n_el_blocks = (f.cfg.layout.n_draw_objects + 255u) / 256u;
    // bit planes: 4 waves x 64 tiles per block and step; sized for the pool (blocks beyond bump.tile exit at once)
    uint32_t n_bit_blocks = (uint32_t)(((uint64_t)f.cfg.tiles_size + 256u * 8u - 1u) / (256u * 8u));
    if (n_bit_blocks > 2048u) n_bit_blocks = 2048u;
    if (n_bit_blocks < 1u) n_bit_blocks = 1u;
    hipLaunchKernelGGL(k_coarse_prep, dim3(n_el_blocks + n_bit_blocks), dim3(256), 0, s, f.cfg, n_el_blocks, f.scene, f.draw_monoids,
                       f.info_bin_data, f.paths, f.tiles, f.bump(), f.coarse_el, f.tile_bits);
    if (mid) (void)hipEventRecord(mid[0], s);
    const uint32_t n_wg = ((wb * hb + 7u) / 8u) * 8u * 4u;
    hipLaunchKernelGGL(k_coarse, dim3(n_wg), dim3(WG), sizeof(CoarseLds), s, f.cfg, f.scene, f.bin_headers, f.info_bin_data, f.coarse_el, f.tile_bits,
                       f.tiles, f.bump(), f.ptcl, !f.no_cull, f.control->work_count, f.tile_order,
                       f.slice_items, f.slice_counters, f.slice_cap, f.cov_cap, f.slice_fills, f.slice_min_fills);
}

}  // namespace vk
