// The bodies of binning and tile_alloc (binning.hip has the head comment and the launches): a header because k_front (flatten.hip)
// runs the same workgroups as the last stages of its one launch.
#pragma once
#include "engine.h"

namespace vk {

constexpr uint32_t N_SLICE = 8;  // 256 draw objects / 32 bits

// 256 draw objects x up to 256 bins per pass.  Coverage bitmaps live in LDS exactly as in the
// reference (8 slices x 256 bins of u32); ranks come from popcounts.  Allocation differs: instead
// of one global atomicAdd(bump.binning) per (workgroup, bin) the per-bin counts are scanned across
// the workgroup and ONE atomic reserves the whole block (same-address atomics are ~12 ns each on
// MI355X).  bin_data chunks of one workgroup are therefore contiguous; consumers only follow
// BinHeader.chunk_offset, so the layout stays valid.
// (`block`: the workgroup's 256 draw objects -- a workgroup of k_binning, or a turn of a workgroup of k_front, flatten.hip)
__device__ __forceinline__ void binning_workgroup(const Config &cfg, uint32_t block, const DrawMonoid *draw_monoids, const PathBbox *path_bbox_buf,
                                                  const Bbox4 *clip_bbox_buf, Bbox4 *intersected_bbox, Bump *bump, uint32_t *bin_data,
                                                  BinHeader *bin_header) {
    __shared__ uint32_t sh_bitmaps[N_SLICE][N_TILE];
    __shared__ uint32_t sh_count[4][N_TILE];  // packed lo/hi partial counts, binning.wgsl:126-133
    __shared__ uint32_t sh_chunk_offset[N_TILE];
    __shared__ uint32_t sh_scan[4];
    __shared__ uint32_t sh_base;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = 0; i < N_SLICE; i++) sh_bitmaps[i][tid] = 0u;
    // binning.wgsl:64-75: flatten overflow is detected here
    if (bump->lines > cfg.lines_size) {
        if (block == 0 && tid == 0) atomicOr(&bump->failed, STAGE_FLATTEN);
        return;
    }
    __syncthreads();
    const uint32_t element_ix = block * 256u + tid;
    const float SX = 1.0f / (float)(N_TILE_X * TILE_WIDTH);
    const float SY = 1.0f / (float)(N_TILE_Y * TILE_HEIGHT);
    int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (element_ix < cfg.layout.n_draw_objects) {
        DrawMonoid dm = draw_monoids[element_ix];
        Bbox4 clip_bbox = {-1e9f, -1e9f, 1e9f, 1e9f};
        if (dm.clip_ix > 0u) clip_bbox = clip_bbox_buf[minu(dm.clip_ix - 1u, cfg.layout.n_clips - 1u)];
        PathBbox pb = path_bbox_buf[dm.path_ix];
        Bbox4 bbox = {maxf(clip_bbox.x0, (float)pb.x0), maxf(clip_bbox.y0, (float)pb.y0), minf(clip_bbox.x1, (float)pb.x1),
                      minf(clip_bbox.y1, (float)pb.y1)};
        intersected_bbox[element_ix] = bbox;
        if (bbox.x0 < bbox.x1 && bbox.y0 < bbox.y1) {
            x0 = f2i(floorf(bbox.x0 * SX));
            y0 = f2i(floorf(bbox.y0 * SY));
            x1 = f2i(ceilf(bbox.x1 * SX));
            y1 = f2i(ceilf(bbox.y1 * SY));
        }
    }
    const int32_t width_in_bins = (int32_t)((cfg.width_in_tiles + N_TILE_X - 1u) / N_TILE_X);
    const int32_t height_in_bins = (int32_t)((cfg.height_in_tiles + N_TILE_Y - 1u) / N_TILE_Y);
    const uint32_t n_bins = (uint32_t)(width_in_bins * height_in_bins);
    const uint32_t aligned_n_bins = (n_bins + N_TILE - 1u) & ~(N_TILE - 1u);
    x0 = clampi(x0, 0, width_in_bins);
    y0 = clampi(y0, 0, height_in_bins);
    x1 = clampi(x1, 0, width_in_bins);
    y1 = clampi(y1, 0, height_in_bins);
    if (x0 == x1) y1 = y0;
    const int32_t y0_width = y0 * width_in_bins, y1_width = y1 * width_in_bins;
    const uint32_t my_slice = tid / 32u;
    const uint32_t my_mask = 1u << (tid & 31u);

    uint32_t next_block = N_TILE;
    for (uint32_t block_start = 0; block_start < n_bins;) {
        for (int32_t y_offset = y0_width; y_offset < y1_width; y_offset += width_in_bins) {
            uint32_t start_bin = maxu((uint32_t)(y_offset + x0), block_start);
            uint32_t end_bin = minu((uint32_t)(y_offset + x1), next_block);
            for (uint32_t bin_ix = start_bin; bin_ix < end_bin; bin_ix++) atomicOr(&sh_bitmaps[my_slice][bin_ix - block_start], my_mask);
        }
        __syncthreads();
        const uint32_t cur_bin_ix = block_start + tid;
        uint32_t element_count = 0u;
        for (uint32_t i = 0; i < 4u; i++) {
            element_count += __popc(sh_bitmaps[i * 2u][tid]);
            uint32_t lo = element_count;
            element_count += __popc(sh_bitmaps[i * 2u + 1u][tid]);
            uint32_t hi = element_count;
            sh_count[i][tid] = lo | (hi << 16);
        }
        uint32_t total;
        uint32_t incl = block256_incl_scan_u32(element_count, sh_scan, &total);
        if (tid == 0u) {
            uint32_t base = total ? atomicAdd(&bump->binning, total) : 0u;
            if (base + total > cfg.binning_size) {
                base = ~0u;  // marks failure for this block
                atomicOr(&bump->failed, STAGE_BINNING);
            }
            sh_base = base;
        }
        __syncthreads();
        const bool ok = sh_base != ~0u;
        uint32_t chunk_offset = ok ? sh_base + (incl - element_count) : 0u;
        sh_chunk_offset[tid] = chunk_offset;
        BinHeader h;
        h.element_count = element_count;
        h.chunk_offset = chunk_offset;
        bin_header[block * aligned_n_bins + cur_bin_ix] = h;
        __syncthreads();
        if (ok) {
            for (int32_t y_offset = y0_width; y_offset < y1_width; y_offset += width_in_bins) {
                uint32_t start_bin = maxu((uint32_t)(y_offset + x0), block_start);
                uint32_t end_bin = minu((uint32_t)(y_offset + x1), next_block);
                for (uint32_t bin_ix = start_bin; bin_ix < end_bin; bin_ix++) {
                    uint32_t sh_bin_ix = bin_ix - block_start;
                    uint32_t out_mask = sh_bitmaps[my_slice][sh_bin_ix];
                    uint32_t idx = __popc(out_mask & (my_mask - 1u));
                    if (my_slice > 0u) {
                        uint32_t count_ix = my_slice - 1u;
                        uint32_t count_packed = sh_count[count_ix / 2u][sh_bin_ix];
                        idx += (count_packed >> (16u * (count_ix & 1u))) & 0xffffu;
                    }
                    bin_data[cfg.layout.bin_data_start + sh_chunk_offset[sh_bin_ix] + idx] = element_ix;
                }
            }
        }
        block_start = next_block;
        if (next_block < aligned_n_bins) {
            __syncthreads();
            for (uint32_t i = 0; i < N_SLICE; i++) sh_bitmaps[i][tid] = 0u;
            __syncthreads();
            next_block += N_TILE;
        }
    }
}

// tile_alloc.wgsl:35-123: per-path tile rectangles, one bump allocation per workgroup, zero fill.
__device__ __forceinline__ void tile_alloc_workgroup(const Config &cfg, uint32_t block, const uint32_t *scene, const Bbox4 *draw_bboxes, Bump *bump,
                                                     Path *paths, Tile *tiles) {
    __shared__ uint32_t sh_scan[4];
    __shared__ uint32_t sh_offset;
    __shared__ uint32_t sh_fill;
    const uint32_t tid = threadIdx.x;
    if ((bump->failed & (STAGE_BINNING | STAGE_FLATTEN | FAILED_SCENE)) != 0u) return;
    const float SX = 1.0f / (float)TILE_WIDTH, SY = 1.0f / (float)TILE_HEIGHT;
    const uint32_t drawobj_ix = block * 256u + tid;
    uint32_t drawtag = DRAWTAG_NOP;
    if (drawobj_ix < cfg.layout.n_draw_objects) drawtag = scene[cfg.layout.draw_tag_base + drawobj_ix];
    int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (drawtag != DRAWTAG_NOP && drawtag != DRAWTAG_END_CLIP) {
        Bbox4 bbox = draw_bboxes[drawobj_ix];
        if (bbox.x0 < bbox.x1 && bbox.y0 < bbox.y1) {
            x0 = f2i(floorf(bbox.x0 * SX));
            y0 = f2i(floorf(bbox.y0 * SY));
            x1 = f2i(ceilf(bbox.x1 * SX));
            y1 = f2i(ceilf(bbox.y1 * SY));
        }
    }
    uint32_t ux0 = (uint32_t)clampi(x0, 0, (int32_t)cfg.width_in_tiles);
    uint32_t uy0 = (uint32_t)clampi(y0, 0, (int32_t)cfg.height_in_tiles);
    uint32_t ux1 = (uint32_t)clampi(x1, 0, (int32_t)cfg.width_in_tiles);
    uint32_t uy1 = (uint32_t)clampi(y1, 0, (int32_t)cfg.height_in_tiles);
    uint32_t tile_count = (ux1 - ux0) * (uy1 - uy0);
    uint32_t total;
    uint32_t incl = block256_incl_scan_u32(tile_count, sh_scan, &total);
    if (tid == 0u) {
        uint32_t offset = total ? atomicAdd(&bump->tile, total) : 0u;
        if (offset + total > cfg.tiles_size) {
            offset = 0u;
            atomicOr(&bump->failed, STAGE_TILE_ALLOC);
            total = 0u;
        }
        sh_offset = offset;
        sh_fill = total;  // zero-fill extent (0 when the allocation failed)
    }
    __syncthreads();
    const uint32_t tile_offset = sh_offset;
    const uint32_t fill = sh_fill;
    if (drawobj_ix < cfg.layout.n_draw_objects) {
        Path p;
        p.bbox[0] = ux0; p.bbox[1] = uy0; p.bbox[2] = ux1; p.bbox[3] = uy1;
        p.tiles = tile_offset + (incl - tile_count);
        p.pad[0] = 0u; p.pad[1] = 0u; p.pad[2] = 0u;
        paths[drawobj_ix] = p;
    }
    // 16-byte stores over the 16-byte aligned middle of the range, single tiles at its ends
    {
        const uint32_t head = minu(fill, tile_offset & 1u);  // (a Tile is 8 bytes: the pool is 16-byte aligned at even tiles)
        unsigned long long *t64 = reinterpret_cast<unsigned long long *>(tiles + tile_offset);
        if (tid < head) t64[tid] = 0ull;
        const uint32_t pairs = (fill - head) / 2u;
        uint4 *t128 = reinterpret_cast<uint4 *>(tiles + tile_offset + head);
        for (uint32_t i = tid; i < pairs; i += 256u) t128[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid == 0u && head + 2u * pairs < fill) t64[fill - 1u] = 0ull;
    }
}

}  // namespace vk
