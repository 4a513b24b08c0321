// coarse: per-tile command lists (PTCL) in draw order.
// Reference: vello_shaders/shader/coarse.wgsl:62-471 (vello/src/render.rs:470-484), PTCL format
// shared/ptcl.wgsl:6-111; CPU twin cpu/coarse.rs.
//
// The reference runs one workgroup per 16x16-tile bin, one thread per tile; each batch of 256 draw objects of the
// bin is merged from the bin lists, a coverage pass reads the Tile record of EVERY (draw object, tile-in-bbox) pair
// into LDS bitmaps, and every thread then walks its tile's bits in draw order, bumping `bump.segments` once per
// (tile, path) and `bump.ptcl` once per chunk.  On MI355X that shape is a few dozen workgroups (49 at 1600^2) of
// dependent global-memory round trips on a 256-CU part, and the coverage pass reads the whole tile pool (a road map's
// long thin paths allocate 10 M tiles of which a third hold anything).  What the stage computes is kept word for
// word (per-tile draw order, the clip_zero_depth state machine, chunked PTCL with CMD_JUMP); how is new:
//
//  * k_coarse_prep (one launch, two jobs by block range): (a) per draw object, everything coarse needs from five
//    buffers (tag, draw flags, first draw-data word, offsets, the Path record) gathered ONCE into a 32-byte record --
//    the reference re-gathers it per (bin, draw object); (b) per 64 tiles of the pool, three bit planes by wave
//    ballot: has segments / backdrop == 0 / backdrop even -- all the coverage test needs (coarse.wgsl:318-341).
//  * k_coarse: one WAVE (= one workgroup) per 8x8-tile quadrant of a bin: 4x the workgroups, no s_barrier, and a
//    draw object is only considered by the quadrants its bbox touches.  Lanes first act as draw objects: the bin's list
//    is streamed 256 entries at a time (entries prefetched one round ahead), survivors are queued in LDS and each
//    gets its 64-bit coverage mask over the quadrant from <= 8 row windows of the bit planes (no Tile reads, no LDS
//    atomics).  Then lanes act as tiles: per batch of <= 256 queued objects the masks are transposed into per-tile
//    bitmaps, the Tile records of the INCLUDED pairs only are staged in LDS (8 loads in flight per lane), and the
//    SIMULATE / EMIT double walk runs from LDS: SIMULATE totals the segments and PTCL chunks each tile needs, a shuffle
//    scan turns them into offsets behind ONE atomic per counter and batch, EMIT writes.
//  * occlusion culling (not in the reference): when a batch holds, for a tile, a fully covering OPAQUE solid-colour
//    draw (CMD_SOLID + CMD_COLOR with alpha 255) outside any clip/blend layer, premultiplied src-over makes everything
//    underneath irrelevant bit-exactly (x*0 + c == c), so the tile's list is restarted at that draw and the covered
//    draws of the batch are never emitted nor given segments.  Only for scenes without clips (the clip state machine
//    must otherwise see every draw); VELLO_HIP_DEBUG_NO_CULL turns it off for word-exact PTCL / segment diffs.
#include "engine.h"

namespace vk {

namespace {

constexpr uint32_t SUB_W = 8;        // a workgroup owns an 8x8-tile quadrant of a bin
constexpr uint32_t NB = 256;         // draw objects per batch
constexpr uint32_t N_SLICE = NB / 32;
constexpr uint32_t RPL = 4;          // bin-list entries examined per lane and round
constexpr uint32_t QCAP = 512;       // queue slots: NB - 1 left over + 64 * RPL new ones; power of two
constexpr uint32_t KMAX = 32;        // Tile records staged in LDS per tile and batch (the rest is read from global memory)
constexpr uint32_t PART_CHUNK = 256; // bin headers (partitions of 256 draw objects) merged at a time
constexpr uint32_t NONE = 0xffffffffu;

struct TileState {
    uint32_t cmd_offset, cmd_limit;
    uint32_t clip_zero_depth, clip_depth, render_blend_depth, max_blend_depth;
};
struct Alloc {
    uint32_t seg_next;    // EMIT: next segment index; SIM: running total
    uint32_t chunk_next;  // EMIT: next PTCL chunk word offset (relative to ptcl_dyn_start); SIM: chunk count
};

template <bool EMIT>
__device__ __forceinline__ void ptcl_store(uint32_t *ptcl, const Config &cfg, uint32_t ix, uint32_t v) {
    if constexpr (EMIT) {
        if (ix < cfg.ptcl_size) ptcl[ix] = v;
    }
}

// One command = one store instruction: the words of a command are consecutive, so a wave's 64 tiles cost 64 cache
// lines per COMMAND instead of per WORD (PTCL offsets are only 4-byte aligned; global memory takes unaligned vectors).
struct __attribute__((packed, aligned(4))) PtclWords2 { uint32_t a, b; };
struct __attribute__((packed, aligned(4))) PtclWords3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(4))) PtclWords4 { uint32_t a, b, c, d; };
template <bool EMIT>
__device__ __forceinline__ void ptcl_store2(uint32_t *ptcl, const Config &cfg, uint32_t ix, uint32_t a, uint32_t b) {
    if constexpr (EMIT) {
        if (ix + 1u < cfg.ptcl_size) {
            *reinterpret_cast<PtclWords2 *>(ptcl + ix) = PtclWords2{a, b};
        } else {
            ptcl_store<true>(ptcl, cfg, ix, a);
            ptcl_store<true>(ptcl, cfg, ix + 1u, b);
        }
    }
}
template <bool EMIT>
__device__ __forceinline__ void ptcl_store3(uint32_t *ptcl, const Config &cfg, uint32_t ix, uint32_t a, uint32_t b, uint32_t c) {
    if constexpr (EMIT) {
        if (ix + 2u < cfg.ptcl_size) {
            *reinterpret_cast<PtclWords3 *>(ptcl + ix) = PtclWords3{a, b, c};
        } else {
            ptcl_store<true>(ptcl, cfg, ix, a);
            ptcl_store<true>(ptcl, cfg, ix + 1u, b);
            ptcl_store<true>(ptcl, cfg, ix + 2u, c);
        }
    }
}
template <bool EMIT>
__device__ __forceinline__ void ptcl_store4(uint32_t *ptcl, const Config &cfg, uint32_t ix, uint32_t a, uint32_t b, uint32_t c,
                                            uint32_t d) {
    if constexpr (EMIT) {
        if (ix + 3u < cfg.ptcl_size) {
            *reinterpret_cast<PtclWords4 *>(ptcl + ix) = PtclWords4{a, b, c, d};
        } else {
            ptcl_store<true>(ptcl, cfg, ix, a);
            ptcl_store<true>(ptcl, cfg, ix + 1u, b);
            ptcl_store<true>(ptcl, cfg, ix + 2u, c);
            ptcl_store<true>(ptcl, cfg, ix + 3u, d);
        }
    }
}

// coarse.wgsl:68-86
template <bool EMIT>
__device__ __forceinline__ void alloc_cmd(TileState &st, Alloc &al, uint32_t size, const Config &cfg, Bump *bump, uint32_t *ptcl) {
    if (st.cmd_offset + size >= st.cmd_limit) {
        if constexpr (EMIT) {
            uint32_t ptcl_dyn_start = cfg.width_in_tiles * cfg.height_in_tiles * PTCL_INITIAL_ALLOC;
            uint32_t new_cmd = ptcl_dyn_start + al.chunk_next;
            al.chunk_next += PTCL_INCREMENT;
            if (new_cmd + PTCL_INCREMENT > cfg.ptcl_size) {
                new_cmd = 0u;
                atomicOr(&bump->failed, STAGE_COARSE);
            }
            ptcl_store2<true>(ptcl, cfg, st.cmd_offset, CMD_JUMP, new_cmd);
            st.cmd_offset = new_cmd;
            st.cmd_limit = new_cmd + (PTCL_INCREMENT - PTCL_HEADROOM);
        } else {
            al.chunk_next += 1u;
            st.cmd_offset = 0u;
            st.cmd_limit = PTCL_INCREMENT - PTCL_HEADROOM;
        }
    }
}

// coarse.wgsl:88-110
template <bool EMIT>
__device__ __forceinline__ void write_path(TileState &st, Alloc &al, Tile tile, uint32_t tile_ix, uint32_t draw_flags, const Config &cfg,
                                           Bump *bump, uint32_t *ptcl, Tile *tiles) {
    uint32_t n_segs = tile.segment_count_or_ix;
    if (n_segs != 0u) {
        uint32_t seg_ix = al.seg_next;
        al.seg_next += n_segs;
        if constexpr (EMIT) tiles[tile_ix].segment_count_or_ix = ~seg_ix;
        alloc_cmd<EMIT>(st, al, 4u, cfg, bump, ptcl);
        uint32_t even_odd = (draw_flags & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u ? 1u : 0u;
        ptcl_store4<EMIT>(ptcl, cfg, st.cmd_offset, CMD_FILL, (n_segs << 1) | even_odd, seg_ix, (uint32_t)tile.backdrop);
        st.cmd_offset += 4u;
    } else {
        alloc_cmd<EMIT>(st, al, 1u, cfg, bump, ptcl);
        ptcl_store<EMIT>(ptcl, cfg, st.cmd_offset, CMD_SOLID);
        st.cmd_offset += 1u;
    }
}

template <bool EMIT>
__device__ __forceinline__ void write2(TileState &st, Alloc &al, uint32_t a, uint32_t b, const Config &cfg, Bump *bump, uint32_t *ptcl) {
    alloc_cmd<EMIT>(st, al, 2u, cfg, bump, ptcl);
    ptcl_store2<EMIT>(ptcl, cfg, st.cmd_offset, a, b);
    st.cmd_offset += 2u;
}
template <bool EMIT>
__device__ __forceinline__ void write3(TileState &st, Alloc &al, uint32_t a, uint32_t b, uint32_t c, const Config &cfg, Bump *bump,
                                       uint32_t *ptcl) {
    alloc_cmd<EMIT>(st, al, 3u, cfg, bump, ptcl);
    ptcl_store3<EMIT>(ptcl, cfg, st.cmd_offset, a, b, c);
    st.cmd_offset += 3u;
}

// Queue of the draw objects that touch this quadrant, in draw order (a ring of QCAP slots).
struct CoarseLds {
    uint32_t tag[QCAP];
    uint32_t flags[QCAP];   // draw_flags = info[di]
    uint32_t w0[QCAP];      // scene[dd]: colour / gradient index / blend mode
    uint32_t dd[QCAP];
    uint32_t di[QCAP];
    uint32_t base[QCAP];    // Tile index of quadrant-local (0, 0) in the path's tile rectangle (may lie outside it)
    uint32_t stride[QCAP];
    uint32_t rect[QCAP];    // x0 | y0 << 4 | w << 8 | h << 12, quadrant-local
    uint32_t cover[QCAP][4];  // 64-bit masks over the quadrant (bit = y * 8 + x): [0..1] tile included, [2..3] tile occluder
    uint32_t bitmaps[N_SLICE][64];  // per tile: which objects of the batch it includes
    Tile rec[KMAX][64];             // per tile: the Tile records of its first KMAX included objects of the batch
    uint32_t part_end[PART_CHUNK];  // inclusive prefix of the element counts of the merged bin headers
    uint32_t part_off[PART_CHUNK];
};

// Iterator over the set bits (draw objects, in order) of this lane's tile.
struct BitIter {
    uint32_t slice, bits;
    __device__ __forceinline__ void init(const CoarseLds &sh, uint32_t lane, uint32_t first_el) {
        slice = first_el / 32u;
        bits = sh.bitmaps[slice][lane] & ~((1u << (first_el & 31u)) - 1u);
    }
    __device__ __forceinline__ uint32_t next(const CoarseLds &sh, uint32_t lane, uint32_t n_slices) {
        while (bits == 0u) {
            if (slice + 1u >= n_slices) return NONE;  // (stays exhausted: bits is 0 and slice is the last one)
            slice += 1u;
            bits = sh.bitmaps[slice][lane];
        }
        uint32_t e = slice * 32u + (uint32_t)(__ffs((int)bits) - 1);
        bits &= bits - 1u;
        return e;
    }
};

// One batch of queued draw objects for this lane's tile: coarse.wgsl:349-452.
// `first_el` / `has_kill`: objects before `first_el` are occluded by the opaque solid draw `first_el`.
template <bool EMIT>
__device__ void process_batch(TileState &st, Alloc &al, const CoarseLds &sh, uint32_t qh, uint32_t n_slices, uint32_t lane, uint32_t tile_x,
                              uint32_t tile_y, uint32_t first_el, bool has_kill, uint32_t list_start, const Config &cfg,
                              const uint32_t *__restrict__ scene, Tile *tiles, Bump *bump, uint32_t *ptcl) {
    BitIter it;
    it.init(sh, lane, first_el);
    uint32_t k = 0u;  // ordinal of the object among this tile's included ones (index of its staged Tile record)
    for (uint32_t el_ix = it.next(sh, lane, n_slices); el_ix != NONE; el_ix = it.next(sh, lane, n_slices), k++) {
        const uint32_t q = (qh + el_ix) & (QCAP - 1u);
        const uint32_t tile_ix = sh.base[q] + sh.stride[q] * tile_y + tile_x;
        const Tile tile = k < KMAX ? sh.rec[k][lane] : tiles[tile_ix];
        const uint32_t drawtag = sh.tag[q];
        if (st.clip_zero_depth == 0u) {
            const uint32_t di = sh.di[q];
            const uint32_t draw_flags = sh.flags[q];
            if (has_kill && el_ix == first_el) {
                // everything emitted so far for this tile is covered: restart the list here
                st.cmd_offset = list_start;
                st.cmd_limit = list_start - 1u + (PTCL_INITIAL_ALLOC - PTCL_HEADROOM);
            }
            switch (drawtag) {
            case DRAWTAG_FILL_COLOR:
                write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                write2<EMIT>(st, al, CMD_COLOR, sh.w0[q], cfg, bump, ptcl);
                break;
            case DRAWTAG_BLURRED_ROUNDED_RECT:
                write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                write3<EMIT>(st, al, CMD_BLUR_RECT, di + 1u, sh.w0[q], cfg, bump, ptcl);
                break;
            case DRAWTAG_FILL_LIN_GRADIENT:
                write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                write3<EMIT>(st, al, CMD_LIN_GRAD, sh.w0[q], di + 1u, cfg, bump, ptcl);
                break;
            case DRAWTAG_FILL_RAD_GRADIENT:
                write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                write3<EMIT>(st, al, CMD_RAD_GRAD, sh.w0[q], di + 1u, cfg, bump, ptcl);
                break;
            case DRAWTAG_FILL_SWEEP_GRADIENT:
                write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                write3<EMIT>(st, al, CMD_SWEEP_GRAD, sh.w0[q], di + 1u, cfg, bump, ptcl);
                break;
            case DRAWTAG_FILL_IMAGE:
                write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                write2<EMIT>(st, al, CMD_IMAGE, di + 1u, cfg, bump, ptcl);
                break;
            case DRAWTAG_BEGIN_CLIP: {
                bool even_odd = (draw_flags & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u;
                int32_t bd = even_odd ? (abs(tile.backdrop) & 1) : tile.backdrop;
                if (tile.segment_count_or_ix == 0u && bd == 0) {
                    st.clip_zero_depth = st.clip_depth + 1u;
                } else {
                    alloc_cmd<EMIT>(st, al, 1u, cfg, bump, ptcl);
                    ptcl_store<EMIT>(ptcl, cfg, st.cmd_offset, CMD_BEGIN_CLIP);
                    st.cmd_offset += 1u;
                    st.render_blend_depth += 1u;
                    st.max_blend_depth = maxu(st.max_blend_depth, st.render_blend_depth);
                }
                st.clip_depth += 1u;
                break;
            }
            case DRAWTAG_END_CLIP:
                st.clip_depth -= 1u;
                write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                write3<EMIT>(st, al, CMD_END_CLIP, sh.w0[q], scene[sh.dd[q] + 1u], cfg, bump, ptcl);
                st.render_blend_depth -= 1u;
                break;
            default: break;
            }
        } else {
            if (drawtag == DRAWTAG_BEGIN_CLIP) {
                st.clip_depth += 1u;
            } else if (drawtag == DRAWTAG_END_CLIP) {
                if (st.clip_depth == st.clip_zero_depth) st.clip_zero_depth = 0u;
                st.clip_depth -= 1u;
            }
        }
    }
}

// 64 bits of a bit plane starting at bit `b` (one 8-byte load at a 4-byte aligned address; the planes carry two words of
// slack behind the last tile).
__device__ __forceinline__ unsigned long long plane_window(const uint32_t *__restrict__ plane, uint32_t b) {
    const PtclWords2 w = *reinterpret_cast<const PtclWords2 *>(plane + (b >> 5));
    return (((unsigned long long)w.b << 32) | (unsigned long long)w.a) >> (b & 31u);
}

}  // namespace

// Job (a), blocks [0, n_el_blocks): the per-draw-object record.  Job (b), the remaining blocks: the tile bit planes.
__global__ void __launch_bounds__(256) k_coarse_prep(Config cfg, uint32_t n_el_blocks, const uint32_t *__restrict__ scene,
                                                     const DrawMonoid *__restrict__ draw_monoids, const uint32_t *__restrict__ info_bin_data,
                                                     const Path *__restrict__ paths, const Tile *__restrict__ tiles, const Bump *__restrict__ bump,
                                                     CoarseEl *__restrict__ coarse_el, uint32_t *__restrict__ tile_bits, uint32_t plane_words) {
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
        if (lane < 6u) {
            const uint32_t p = lane >> 1, h = lane & 1u;
            const unsigned long long m = p == 0u ? ms : (p == 1u ? mz : mo);
            tile_bits[(size_t)p * plane_words + chunk * 2u + h] = (uint32_t)(m >> (32u * h));
        }
    }
}

__global__ void __launch_bounds__(64) k_coarse(Config cfg, const uint32_t *__restrict__ scene, const BinHeader *__restrict__ bin_headers,
                                               const uint32_t *__restrict__ info_bin_data, const CoarseEl *__restrict__ coarse_el,
                                               const uint32_t *__restrict__ tile_bits, uint32_t plane_words, Tile *tiles, Bump *bump,
                                               uint32_t *ptcl, bool allow_cull) {
    __shared__ CoarseLds sh;
    const uint32_t lane = threadIdx.x;
    {  // coarse.wgsl:161-176
        uint32_t failed = bump->failed & (STAGE_BINNING | STAGE_TILE_ALLOC | STAGE_FLATTEN | FAILED_SCENE);
        if (bump->seg_counts > cfg.seg_counts_size) failed |= STAGE_PATH_COUNT;
        if (failed != 0u) {
            if (blockIdx.x == 0 && lane == 0) atomicOr(&bump->failed, failed);
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
    const uint32_t tile_x = lane % SUB_W;
    const uint32_t tile_y = lane / SUB_W;
    const uint32_t this_tile_ix = (sub_y0 + tile_y) * cfg.width_in_tiles + sub_x0 + tile_x;
    const uint32_t *plane_s = tile_bits;
    const uint32_t *plane_z = tile_bits + plane_words;
    const uint32_t *plane_o = tile_bits + 2u * (size_t)plane_words;

    TileState st;
    st.cmd_offset = this_tile_ix * PTCL_INITIAL_ALLOC;
    st.cmd_limit = st.cmd_offset + (PTCL_INITIAL_ALLOC - PTCL_HEADROOM);
    st.clip_zero_depth = 0u; st.clip_depth = 0u; st.render_blend_depth = 0u; st.max_blend_depth = 0u;
    const uint32_t blend_offset = st.cmd_offset;
    st.cmd_offset += 1u;
    const uint32_t list_start = st.cmd_offset;
    const bool cull = allow_cull && cfg.layout.n_clips == 0u;

    // ---- the bin's element stream: bin headers merged PART_CHUNK partitions at a time (coarse.wgsl:218-263) ----
    uint32_t partition_ix = 0u;                 // next partition to merge
    uint32_t chunk_total = 0u, chunk_pos = 0u;  // elements of the merged partitions / handed out so far
    auto refill = [&]() {
        wave_lds_sync();  // the binary searches of the previous round are done with part_end / part_off
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
        chunk_total = carry;
        chunk_pos = 0u;
        partition_ix += PART_CHUNK;
        wave_lds_sync();
    };
    // draw object indices of the next round (NONE = no entry); advances the stream
    auto fetch_indices = [&](uint32_t (&d)[RPL]) {
        while (chunk_pos >= chunk_total && partition_ix < n_partitions) refill();
#pragma unroll
        for (uint32_t j = 0; j < RPL; j++) {
            const uint32_t ix = chunk_pos + j * 64u + lane;
            d[j] = NONE;
            if (ix < chunk_total) {
                uint32_t part = 0u;
#pragma unroll
                for (uint32_t i = 0; i < 8u; i++) {
                    const uint32_t probe = part + (128u >> i);
                    if (ix >= sh.part_end[probe - 1u]) part = probe;
                }
                const uint32_t local = ix - (part > 0u ? sh.part_end[part - 1u] : 0u);
                d[j] = info_bin_data[cfg.layout.bin_data_start + sh.part_off[part] + local];
            }
        }
        chunk_pos = minu(chunk_pos + 64u * RPL, chunk_total);
    };

    uint32_t qh = 0u, qlen = 0u;  // queue head (ring index) and length
    uint32_t d_cur[RPL];
    fetch_indices(d_cur);
    bool more = true;
    while (more) {
        // ---- lanes as draw objects: records of this round, indices of the next ----
        CoarseEl el[RPL];
#pragma unroll
        for (uint32_t j = 0; j < RPL; j++) {
            el[j].tag = DRAWTAG_NOP;
            el[j].flags = 0u; el[j].w0 = 0u; el[j].dd = 0u; el[j].di = 0u; el[j].tiles = 0u; el[j].bbox_x = 0u; el[j].bbox_y = 0u;
            if (d_cur[j] != NONE) {
                const uint4 *p = reinterpret_cast<const uint4 *>(coarse_el + d_cur[j]);
                const uint4 a = p[0], b = p[1];
                el[j].tag = a.x; el[j].flags = a.y; el[j].w0 = a.z; el[j].dd = a.w;
                el[j].di = b.x; el[j].tiles = b.y; el[j].bbox_x = b.z; el[j].bbox_y = b.w;
            }
        }
        uint32_t d_next[RPL];
        fetch_indices(d_next);
        bool any_next = false;
#pragma unroll
        for (uint32_t j = 0; j < RPL; j++) any_next = any_next || d_next[j] != NONE;
        more = __ballot(any_next) != 0ull;
        // keep the objects whose tile rectangle meets this quadrant, in order (coarse.wgsl:264-289)
        const uint32_t q_new = qh + qlen;  // (un-wrapped) slot of the first object appended this round
#pragma unroll
        for (uint32_t j = 0; j < RPL; j++) {
            const int32_t bx0 = (int32_t)(el[j].bbox_x & 0xffffu), bx1 = (int32_t)(el[j].bbox_x >> 16);
            const int32_t by0 = (int32_t)(el[j].bbox_y & 0xffffu), by1 = (int32_t)(el[j].bbox_y >> 16);
            const int32_t dx = bx0 - (int32_t)sub_x0, dy = by0 - (int32_t)sub_y0;
            const int32_t x0 = clampi(dx, 0, (int32_t)SUB_W), y0 = clampi(dy, 0, (int32_t)SUB_W);
            const int32_t x1 = clampi(bx1 - (int32_t)sub_x0, 0, (int32_t)SUB_W), y1 = clampi(by1 - (int32_t)sub_y0, 0, (int32_t)SUB_W);
            const bool keep = el[j].tag != DRAWTAG_NOP && x1 > x0 && y1 > y0;
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const uint32_t q = (qh + qlen + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (QCAP - 1u);
                const uint32_t stride = (uint32_t)(bx1 - bx0);
                sh.tag[q] = el[j].tag;
                sh.flags[q] = el[j].flags;
                sh.w0[q] = el[j].w0;
                sh.dd[q] = el[j].dd;
                sh.di[q] = el[j].di;
                sh.stride[q] = stride;
                sh.base[q] = el[j].tiles - (uint32_t)(dy * (int32_t)stride + dx);
                sh.rect[q] = (uint32_t)x0 | ((uint32_t)y0 << 4) | ((uint32_t)(x1 - x0) << 8) | ((uint32_t)(y1 - y0) << 12);
            }
            qlen += (uint32_t)__popcll(m);
        }
        wave_lds_sync();
        // coverage masks of the new objects, one object per lane: <= 8 row windows of the bit planes
        // (coarse.wgsl:290-347 reads a Tile per (object, tile) pair and sets LDS bits with atomics)
        const uint32_t q_end = qh + qlen;
        for (uint32_t qi = q_new + lane; qi < q_end; qi += 64u) {
            const uint32_t q = qi & (QCAP - 1u);
            const uint32_t rect = sh.rect[q], tag = sh.tag[q], w0 = sh.w0[q];
            const uint32_t x0 = rect & 15u, y0 = (rect >> 4) & 15u, w = (rect >> 8) & 15u, h = rect >> 12;
            const uint32_t base = sh.base[q], stride = sh.stride[q];
            const bool is_clip = (tag & 1u) != 0u;
            const uint32_t BLEND_CLIP = (128u << 8) | 3u;
            const bool is_blend = is_clip && w0 != BLEND_CLIP;
            const bool even_odd = (sh.flags[q] & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u;
            const uint32_t *plane_c = even_odd ? plane_o : plane_z;
            const uint32_t wmask = (1u << w) - 1u;
            unsigned long long ws[SUB_W], wc[SUB_W];
#pragma unroll
            for (uint32_t r = 0; r < SUB_W; r++) {
                ws[r] = 0ull;
                wc[r] = 0ull;
                if (r < h) {
                    const uint32_t b = base + stride * (y0 + r) + x0;
                    ws[r] = plane_window(plane_s, b);
                    wc[r] = plane_window(plane_c, b);
                }
            }
            unsigned long long inc = 0ull, kil = 0ull;
#pragma unroll
            for (uint32_t r = 0; r < SUB_W; r++) {
                if (r < h) {
                    const uint32_t s = (uint32_t)ws[r] & wmask;
                    const uint32_t clear = (uint32_t)wc[r] & wmask;  // backdrop_clear per tile of the row
                    // include_tile = n_segs != 0 || (backdrop_clear == is_clip) || is_blend
                    const uint32_t in = is_blend ? wmask : (s | ((is_clip ? clear : ~clear) & wmask));
                    const uint32_t shift = (y0 + r) * SUB_W + x0;
                    inc |= (unsigned long long)in << shift;
                    kil |= (unsigned long long)(in & ~s) << shift;
                }
            }
            // fully covering opaque solid colour: occludes every earlier draw of the tile
            if (!(cull && tag == DRAWTAG_FILL_COLOR && (w0 >> 24) == 0xffu)) kil = 0ull;
            sh.cover[q][0] = (uint32_t)inc;
            sh.cover[q][1] = (uint32_t)(inc >> 32);
            sh.cover[q][2] = (uint32_t)kil;
            sh.cover[q][3] = (uint32_t)(kil >> 32);
        }
        wave_lds_sync();

        // ---- lanes as tiles: full batches, and whatever is left when the stream has ended ----
        while (qlen >= NB || (!more && qlen > 0u)) {
            const uint32_t n = minu(qlen, NB);
            const uint32_t n_slices = (n + 31u) / 32u;
            // transpose the objects' masks into this tile's bitmap; remember the last occluder
            uint32_t first_el = 0u;
            bool has_kill = false;
            const bool hi = lane >= 32u;
            const uint32_t sh_l = lane & 31u;
            for (uint32_t sl = 0; sl < n_slices; sl++) {
                uint32_t word = 0u;
                const uint32_t cnt = minu(32u, n - sl * 32u);
                for (uint32_t b = 0; b < cnt; b++) {
                    const uint32_t q = (qh + sl * 32u + b) & (QCAP - 1u);
                    const uint32_t mi = ((hi ? sh.cover[q][1] : sh.cover[q][0]) >> sh_l) & 1u;
                    const uint32_t mk = ((hi ? sh.cover[q][3] : sh.cover[q][2]) >> sh_l) & 1u;
                    word |= mi << b;
                    if (mk != 0u) {
                        first_el = sl * 32u + b;
                        has_kill = true;
                    }
                }
                sh.bitmaps[sl][lane] = word;
            }
            wave_lds_sync();
            // stage the Tile records of this tile's included objects (from the occluder on), 8 loads in flight
            {
                BitIter it;
                it.init(sh, lane, first_el);
                bool done = false;
                for (uint32_t k0 = 0; k0 < KMAX && !done; k0 += 8u) {
                    Tile t[8];
                    bool valid[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8u; u++) {
                        const uint32_t e = it.next(sh, lane, n_slices);
                        valid[u] = e != NONE;
                        t[u] = Tile{0, 0u};
                        if (valid[u]) {
                            const uint32_t q = (qh + e) & (QCAP - 1u);
                            t[u] = tiles[sh.base[q] + sh.stride[q] * tile_y + tile_x];
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 8u; u++)
                        if (valid[u]) sh.rec[k0 + u][lane] = t[u];
                    done = !valid[7];
                }
            }
            wave_lds_sync();
            // SIMULATE: how many segments / PTCL chunks does this tile need for the batch?
            TileState sim = st;
            Alloc cnt;
            cnt.seg_next = 0u;
            cnt.chunk_next = 0u;
            process_batch<false>(sim, cnt, sh, qh, n_slices, lane, tile_x, tile_y, first_el, has_kill, list_start, cfg, scene, tiles, bump,
                                 ptcl);
            const uint32_t seg_incl = wave_incl_scan_u32(cnt.seg_next, (int)lane);
            const uint32_t chunk_incl = wave_incl_scan_u32(cnt.chunk_next, (int)lane);
            const uint32_t total_segs = (uint32_t)__shfl((int)seg_incl, 63);
            const uint32_t total_chunks = (uint32_t)__shfl((int)chunk_incl, 63);
            uint32_t seg_base = 0u, chunk_base = 0u;
            if (lane == 0u) {
                seg_base = total_segs ? atomicAdd(&bump->segments, total_segs) : 0u;
                chunk_base = total_chunks ? atomicAdd(&bump->ptcl, total_chunks * PTCL_INCREMENT) : 0u;
            }
            seg_base = (uint32_t)__shfl((int)seg_base, 0);
            chunk_base = (uint32_t)__shfl((int)chunk_base, 0);
            // EMIT
            Alloc al;
            al.seg_next = seg_base + (seg_incl - cnt.seg_next);
            al.chunk_next = chunk_base + (chunk_incl - cnt.chunk_next) * PTCL_INCREMENT;
            process_batch<true>(st, al, sh, qh, n_slices, lane, tile_x, tile_y, first_el, has_kill, list_start, cfg, scene, tiles, bump, ptcl);
            qh = (qh + n) & (QCAP - 1u);
            qlen -= n;
            wave_lds_sync();  // bitmaps / rec / the freed queue slots are rewritten next
        }
        if (more) {
#pragma unroll
            for (uint32_t j = 0; j < RPL; j++) d_cur[j] = d_next[j];
        }
    }
    if (sub_x0 + tile_x < cfg.width_in_tiles && sub_y0 + tile_y < cfg.height_in_tiles) {
        ptcl_store<true>(ptcl, cfg, st.cmd_offset, CMD_END);
        uint32_t blend_ix = 0u;
        if (st.max_blend_depth > BLEND_STACK_SPLIT) {
            uint32_t scratch_size = (st.max_blend_depth - BLEND_STACK_SPLIT) * TILE_WIDTH * TILE_HEIGHT;
            blend_ix = atomicAdd(&bump->blend, scratch_size);
            if (blend_ix + scratch_size > cfg.blend_size) atomicOr(&bump->failed, STAGE_COARSE);
        }
        ptcl_store<true>(ptcl, cfg, blend_offset, blend_ix);
    }
}

void launch_coarse(const Frame &f, hipStream_t s) {
    const uint32_t wb = (f.cfg.width_in_tiles + 15u) / 16u, hb = (f.cfg.height_in_tiles + 15u) / 16u;
    if (wb * hb == 0) return;
    const uint32_t n_el_blocks = (f.cfg.layout.n_draw_objects + 255u) / 256u;
    // bit planes: 4 waves x 64 tiles per block and step; sized for the pool (blocks beyond bump.tile exit at once)
    uint32_t n_bit_blocks = (uint32_t)(((uint64_t)f.cfg.tiles_size + 256u * 8u - 1u) / (256u * 8u));
    if (n_bit_blocks > 2048u) n_bit_blocks = 2048u;
    if (n_bit_blocks < 1u) n_bit_blocks = 1u;
    hipLaunchKernelGGL(k_coarse_prep, dim3(n_el_blocks + n_bit_blocks), dim3(256), 0, s, f.cfg, n_el_blocks, f.scene, f.draw_monoids,
                       f.info_bin_data, f.paths, f.tiles, f.bump(), f.coarse_el, f.tile_bits, f.tile_bits_plane_words);
    const uint32_t n_wg = ((wb * hb + 7u) / 8u) * 8u * 4u;
    hipLaunchKernelGGL(k_coarse, dim3(n_wg), dim3(64), 0, s, f.cfg, f.scene, f.bin_headers, f.info_bin_data, f.coarse_el, f.tile_bits,
                       f.tile_bits_plane_words, f.tiles, f.bump(), f.ptcl, !f.no_cull);
}

}  // namespace vk
