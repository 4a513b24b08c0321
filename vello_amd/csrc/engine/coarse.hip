// coarse: per-tile command lists (PTCL) in draw order.
// Reference: vello_shaders/shader/coarse.wgsl:62-471 (vello/src/render.rs:470-484), PTCL format
// shared/ptcl.wgsl:6-111; CPU twin cpu/coarse.rs.
//
// Structure follows the reference (one workgroup per 16x16-tile bin, one thread per tile, bin
// partitions merged 256 draw objects at a time through LDS coverage bitmaps) because per-tile draw
// order and the clip_zero_depth state machine depend on it.  gfx950 changes:
//  * Hillis-Steele LDS scans -> wave64 shuffle scans;
//  * the reference bumps `bump.segments` once per (tile, path) and `bump.ptcl` once per PTCL chunk
//    with global atomics (coarse.wgsl:70,92).  Here every batch of 256 draw objects is walked
//    twice: a SIMULATE pass totals the segments and PTCL chunks each tile will need, a workgroup
//    scan turns them into offsets behind ONE atomic per counter, and the EMIT pass writes;
//  * per-element data (tag, draw flags, first draw-data word, offsets) is staged into LDS once per batch
//    instead of being re-read from global memory for every (tile, element) pair;
//  * occlusion culling (not in the reference): when a batch holds, for a tile, a fully covering OPAQUE
//    solid-colour draw (CMD_SOLID + CMD_COLOR with alpha 255) outside any clip/blend layer, premultiplied
//    src-over makes everything underneath irrelevant bit-exactly (x*0 + c == c), so the tile's list is
//    restarted at that draw and the covered draws of the batch are never emitted nor given segments.
//    Enabled only for scenes without clips (the clip state machine must otherwise see every draw).
#include "engine.h"

namespace vk {

namespace {

constexpr uint32_t N_SLICE = 8;

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

struct ElemLds {
    uint32_t tag[256];
    uint32_t flags[256];  // draw_flags = info[di]
    uint32_t w0[256];     // scene[dd]: colour / gradient index / blend mode
    uint32_t dd[256];
    uint32_t di[256];
};

// One batch of (up to) 256 draw objects for this thread's tile: coarse.wgsl:349-452.
// `first_el` / `has_kill`: elements before `first_el` are occluded by the opaque solid draw `first_el`.
template <bool EMIT>
__device__ void process_batch(TileState &st, Alloc &al, const uint32_t (*sh_bitmaps)[N_TILE], const ElemLds &el,
                              const uint32_t *sh_tile_base, const uint32_t *sh_tile_stride, uint32_t tid, uint32_t tile_x,
                              uint32_t tile_y, uint32_t first_el, bool has_kill, uint32_t list_start, const Config &cfg,
                              const uint32_t *__restrict__ scene, Tile *tiles, Bump *bump, uint32_t *ptcl) {
    // Elements of this tile in draw order.  The Tile records of the NEXT elements are requested before the current one
    // is processed: the walk is one dependent global load per element on a workgroup that has nothing else to run.
    uint32_t it_slice = first_el / 32u;
    uint32_t it_bits = sh_bitmaps[it_slice][tid] & ~((1u << (first_el & 31u)) - 1u);
    auto next_element = [&]() -> uint32_t {
        while (it_bits == 0u) {
            it_slice += 1u;
            if (it_slice >= N_SLICE) return 0xffffffffu;
            it_bits = sh_bitmaps[it_slice][tid];
        }
        uint32_t e = it_slice * 32u + (uint32_t)(__ffs((int)it_bits) - 1);
        it_bits &= it_bits - 1u;
        return e;
    };
    // two elements ahead: the processing of one element (~500 cycles) does not cover a Tile fetch (~900)
    uint32_t q_el[2], q_ix[2] = {0u, 0u};
    Tile q_tile[2] = {};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        q_el[k] = next_element();
        if (q_el[k] != 0xffffffffu) {
            q_ix[k] = sh_tile_base[q_el[k]] + sh_tile_stride[q_el[k]] * tile_y + tile_x;
            q_tile[k] = tiles[q_ix[k]];
        }
    }
    while (q_el[0] != 0xffffffffu) {
        {
            const uint32_t el_ix = q_el[0];
            const uint32_t tile_ix = q_ix[0];
            const Tile tile = q_tile[0];
            q_el[0] = q_el[1];
            q_ix[0] = q_ix[1];
            q_tile[0] = q_tile[1];
            q_el[1] = q_el[0] != 0xffffffffu ? next_element() : 0xffffffffu;
            if (q_el[1] != 0xffffffffu) {
                q_ix[1] = sh_tile_base[q_el[1]] + sh_tile_stride[q_el[1]] * tile_y + tile_x;
                q_tile[1] = tiles[q_ix[1]];
            }
            uint32_t drawtag = el.tag[el_ix];
            if (st.clip_zero_depth == 0u) {
                uint32_t dd = el.dd[el_ix];
                uint32_t di = el.di[el_ix];
                uint32_t draw_flags = el.flags[el_ix];
                if (has_kill && el_ix == first_el) {
                    // everything emitted so far for this tile is covered: restart the list here
                    st.cmd_offset = list_start;
                    st.cmd_limit = list_start - 1u + (PTCL_INITIAL_ALLOC - PTCL_HEADROOM);
                }
                switch (drawtag) {
                case DRAWTAG_FILL_COLOR:
                    write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                    write2<EMIT>(st, al, CMD_COLOR, el.w0[el_ix], cfg, bump, ptcl);
                    break;
                case DRAWTAG_BLURRED_ROUNDED_RECT:
                    write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                    write3<EMIT>(st, al, CMD_BLUR_RECT, di + 1u, el.w0[el_ix], cfg, bump, ptcl);
                    break;
                case DRAWTAG_FILL_LIN_GRADIENT:
                    write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                    write3<EMIT>(st, al, CMD_LIN_GRAD, el.w0[el_ix], di + 1u, cfg, bump, ptcl);
                    break;
                case DRAWTAG_FILL_RAD_GRADIENT:
                    write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                    write3<EMIT>(st, al, CMD_RAD_GRAD, el.w0[el_ix], di + 1u, cfg, bump, ptcl);
                    break;
                case DRAWTAG_FILL_SWEEP_GRADIENT:
                    write_path<EMIT>(st, al, tile, tile_ix, draw_flags, cfg, bump, ptcl, tiles);
                    write3<EMIT>(st, al, CMD_SWEEP_GRAD, el.w0[el_ix], di + 1u, cfg, bump, ptcl);
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
                    write3<EMIT>(st, al, CMD_END_CLIP, el.w0[el_ix], scene[dd + 1u], cfg, bump, ptcl);
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
}

}  // namespace

__global__ void __launch_bounds__(256) k_coarse(Config cfg, const uint32_t *__restrict__ scene, const DrawMonoid *__restrict__ draw_monoids,
                                                const BinHeader *__restrict__ bin_headers, const uint32_t *__restrict__ info_bin_data,
                                                const Path *__restrict__ paths, Tile *tiles, Bump *bump, uint32_t *ptcl, bool allow_cull) {
    __shared__ uint32_t sh_bitmaps[N_SLICE][N_TILE];
    __shared__ uint32_t sh_kill[N_SLICE][N_TILE];
    __shared__ ElemLds sh_el;
    __shared__ uint32_t sh_part_count[256];
    __shared__ uint32_t sh_part_offsets[256];
    __shared__ uint32_t sh_drawobj_ix[256];
    __shared__ uint32_t sh_tile_stride[256];
    __shared__ uint32_t sh_tile_width[256];
    __shared__ uint32_t sh_tile_x0y0[256];
    __shared__ uint32_t sh_tile_count[256];
    __shared__ uint32_t sh_tile_base[256];
    __shared__ uint32_t sh_scan[4];
    __shared__ uint32_t sh_seg_base, sh_chunk_base;
    const uint32_t tid = threadIdx.x;
    {  // coarse.wgsl:161-176
        uint32_t failed = bump->failed & (STAGE_BINNING | STAGE_TILE_ALLOC | STAGE_FLATTEN | FAILED_SCENE);
        if (bump->seg_counts > cfg.seg_counts_size) failed |= STAGE_PATH_COUNT;
        if (failed != 0u) {
            if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicOr(&bump->failed, failed);
            return;
        }
    }
    const uint32_t width_in_bins = (cfg.width_in_tiles + N_TILE_X - 1u) / N_TILE_X;
    const uint32_t height_in_bins = (cfg.height_in_tiles + N_TILE_Y - 1u) / N_TILE_Y;
    const uint32_t bin_ix = width_in_bins * blockIdx.y + blockIdx.x;
    const uint32_t aligned_n_bins = (width_in_bins * height_in_bins + N_TILE - 1u) & ~(N_TILE - 1u);
    const uint32_t n_partitions = (cfg.layout.n_draw_objects + N_TILE - 1u) / N_TILE;
    const uint32_t bin_tile_x = N_TILE_X * blockIdx.x;
    const uint32_t bin_tile_y = N_TILE_Y * blockIdx.y;
    const uint32_t tile_x = tid % N_TILE_X;
    const uint32_t tile_y = tid / N_TILE_X;
    const uint32_t this_tile_ix = (bin_tile_y + tile_y) * cfg.width_in_tiles + bin_tile_x + tile_x;

    TileState st;
    st.cmd_offset = this_tile_ix * PTCL_INITIAL_ALLOC;
    st.cmd_limit = st.cmd_offset + (PTCL_INITIAL_ALLOC - PTCL_HEADROOM);
    st.clip_zero_depth = 0u; st.clip_depth = 0u; st.render_blend_depth = 0u; st.max_blend_depth = 0u;
    const uint32_t blend_offset = st.cmd_offset;
    st.cmd_offset += 1u;
    const uint32_t list_start = st.cmd_offset;
    const bool cull = allow_cull && cfg.layout.n_clips == 0u;

    uint32_t partition_ix = 0u, rd_ix = 0u, wr_ix = 0u, part_start_ix = 0u, ready_ix = 0u;

    while (true) {
        for (uint32_t i = 0; i < N_SLICE; i++) {
            sh_bitmaps[i][tid] = 0u;
            sh_kill[i][tid] = 0u;
        }

        // merge the per-partition bin lists of this bin, 256 elements at a time (coarse.wgsl:218-263)
        while (true) {
            if (ready_ix == wr_ix && partition_ix < n_partitions) {
                part_start_ix = ready_ix;
                uint32_t count = 0u;
                if (partition_ix + tid < n_partitions) {
                    BinHeader bh = bin_headers[(partition_ix + tid) * aligned_n_bins + bin_ix];
                    count = bh.element_count;
                    sh_part_offsets[tid] = bh.chunk_offset;
                }
                uint32_t total;
                uint32_t incl = block256_incl_scan_u32(count, sh_scan, &total);
                sh_part_count[tid] = part_start_ix + incl;
                ready_ix = part_start_ix + total;
                partition_ix += 256u;
                __syncthreads();
            }
            uint32_t ix = rd_ix + tid;
            if (ix >= wr_ix && ix < ready_ix) {
                uint32_t part_ix = 0u;
#pragma unroll
                for (uint32_t i = 0; i < 8u; i++) {
                    uint32_t probe = part_ix + (128u >> i);
                    if (ix >= sh_part_count[probe - 1u]) part_ix = probe;
                }
                ix -= part_ix > 0u ? sh_part_count[part_ix - 1u] : part_start_ix;
                uint32_t offset = cfg.layout.bin_data_start + sh_part_offsets[part_ix];
                sh_drawobj_ix[tid] = info_bin_data[offset + ix];
            }
            wr_ix = minu(rd_ix + N_TILE, ready_ix);
            if (wr_ix - rd_ix >= N_TILE || (wr_ix >= ready_ix && partition_ix >= n_partitions)) break;
            __syncthreads();
        }

        // per-element tile rectangles inside this bin (coarse.wgsl:264-289)
        uint32_t tag = DRAWTAG_NOP;
        uint32_t drawobj_ix = 0u;
        if (tid + rd_ix < wr_ix) {
            drawobj_ix = sh_drawobj_ix[tid];
            tag = scene[cfg.layout.draw_tag_base + drawobj_ix];
        }
        uint32_t tile_count = 0u;
        sh_el.tag[tid] = tag;
        if (tag != DRAWTAG_NOP) {
            DrawMonoid dm = draw_monoids[drawobj_ix];
            uint32_t dd = cfg.layout.draw_data_base + dm.scene_offset;
            sh_el.dd[tid] = dd;
            sh_el.di[tid] = dm.info_offset;
            sh_el.flags[tid] = info_bin_data[dm.info_offset];
            sh_el.w0[tid] = scene[dd];
            uint32_t path_ix = dm.path_ix;
            Path path = paths[path_ix];
            uint32_t stride = path.bbox[2] - path.bbox[0];
            sh_tile_stride[tid] = stride;
            int32_t dx = (int32_t)path.bbox[0] - (int32_t)bin_tile_x;
            int32_t dy = (int32_t)path.bbox[1] - (int32_t)bin_tile_y;
            int32_t x0 = clampi(dx, 0, (int32_t)N_TILE_X);
            int32_t y0 = clampi(dy, 0, (int32_t)N_TILE_Y);
            int32_t x1 = clampi((int32_t)path.bbox[2] - (int32_t)bin_tile_x, 0, (int32_t)N_TILE_X);
            int32_t y1 = clampi((int32_t)path.bbox[3] - (int32_t)bin_tile_y, 0, (int32_t)N_TILE_Y);
            sh_tile_width[tid] = (uint32_t)(x1 - x0);
            sh_tile_x0y0[tid] = (uint32_t)x0 | ((uint32_t)y0 << 16);
            tile_count = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
            sh_tile_base[tid] = path.tiles - (uint32_t)(dy * (int32_t)stride + dx);
        }
        uint32_t total_tile_count;
        uint32_t tc_incl = block256_incl_scan_u32(tile_count, sh_scan, &total_tile_count);
        sh_tile_count[tid] = tc_incl;
        __syncthreads();

        // tile x element coverage bitmaps (coarse.wgsl:290-347)
        for (uint32_t ix = tid; ix < total_tile_count; ix += N_TILE) {
            uint32_t el_ix = 0u;
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) {
                uint32_t probe = el_ix + (128u >> i);
                if (ix >= sh_tile_count[probe - 1u]) el_ix = probe;
            }
            uint32_t el_tag = sh_el.tag[el_ix];
            uint32_t seq_ix = ix - (el_ix > 0u ? sh_tile_count[el_ix - 1u] : 0u);
            uint32_t width = sh_tile_width[el_ix];
            uint32_t x0y0 = sh_tile_x0y0[el_ix];
            uint32_t x = (x0y0 & 0xffffu) + seq_ix % width;
            uint32_t y = (x0y0 >> 16) + seq_ix / width;
            uint32_t tile_ix = sh_tile_base[el_ix] + sh_tile_stride[el_ix] * y + x;
            Tile tile = tiles[tile_ix];
            bool is_clip = (el_tag & 1u) != 0u;
            bool is_blend = false;
            if (is_clip) {
                const uint32_t BLEND_CLIP = (128u << 8) | 3u;
                is_blend = sh_el.w0[el_ix] != BLEND_CLIP;
            }
            bool even_odd = (sh_el.flags[el_ix] & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u;
            uint32_t n_segs = tile.segment_count_or_ix;
            int32_t bd = even_odd ? (abs(tile.backdrop) & 1) : tile.backdrop;
            bool backdrop_clear = bd == 0;
            bool include_tile = n_segs != 0u || (backdrop_clear == is_clip) || is_blend;
            if (include_tile) {
                atomicOr(&sh_bitmaps[el_ix / 32u][y * N_TILE_X + x], 1u << (el_ix & 31u));
                // fully covering opaque solid colour: occludes every earlier draw of this tile
                if (cull && el_tag == DRAWTAG_FILL_COLOR && n_segs == 0u && (sh_el.w0[el_ix] >> 24) == 0xffu)
                    atomicOr(&sh_kill[el_ix / 32u][y * N_TILE_X + x], 1u << (el_ix & 31u));
            }
        }
        __syncthreads();

        // last occluder of this tile in the batch (if any): earlier elements are skipped by both passes
        uint32_t first_el = 0u;
        bool has_kill = false;
        for (int sl = (int)N_SLICE - 1; sl >= 0; sl--) {
            uint32_t kb = sh_kill[sl][tid];
            if (kb != 0u) {
                first_el = (uint32_t)sl * 32u + (31u - (uint32_t)__clz((int)kb));
                has_kill = true;
                break;
            }
        }
        // SIMULATE: how many segments / PTCL chunks does this tile need for the batch?
        TileState sim = st;
        Alloc cnt;
        cnt.seg_next = 0u;
        cnt.chunk_next = 0u;
        process_batch<false>(sim, cnt, sh_bitmaps, sh_el, sh_tile_base, sh_tile_stride, tid, tile_x, tile_y, first_el, has_kill,
                             list_start, cfg, scene, tiles, bump, ptcl);
        uint32_t total_segs, total_chunks;
        uint32_t seg_incl = block256_incl_scan_u32(cnt.seg_next, sh_scan, &total_segs);
        uint32_t chunk_incl = block256_incl_scan_u32(cnt.chunk_next, sh_scan, &total_chunks);
        if (tid == 0u) {
            sh_seg_base = total_segs ? atomicAdd(&bump->segments, total_segs) : 0u;
            sh_chunk_base = total_chunks ? atomicAdd(&bump->ptcl, total_chunks * PTCL_INCREMENT) : 0u;
        }
        __syncthreads();
        // EMIT
        Alloc al;
        al.seg_next = sh_seg_base + (seg_incl - cnt.seg_next);
        al.chunk_next = sh_chunk_base + (chunk_incl - cnt.chunk_next) * PTCL_INCREMENT;
        process_batch<true>(st, al, sh_bitmaps, sh_el, sh_tile_base, sh_tile_stride, tid, tile_x, tile_y, first_el, has_kill,
                            list_start, cfg, scene, tiles, bump, ptcl);

        rd_ix += N_TILE;
        if (rd_ix >= ready_ix && partition_ix >= n_partitions) break;
        __syncthreads();
    }
    if (bin_tile_x + tile_x < cfg.width_in_tiles && bin_tile_y + tile_y < cfg.height_in_tiles) {
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
    uint32_t wb = (f.cfg.width_in_tiles + 15u) / 16u, hb = (f.cfg.height_in_tiles + 15u) / 16u;
    if (wb * hb == 0) return;
    hipLaunchKernelGGL(k_coarse, dim3(wb, hb), dim3(256), 0, s, f.cfg, f.scene, f.draw_monoids, f.bin_headers, f.info_bin_data, f.paths,
                       f.tiles, f.bump(), f.ptcl, !f.no_cull);
}

}  // namespace vk
