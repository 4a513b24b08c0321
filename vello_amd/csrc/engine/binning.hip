// binning and tile_alloc.
// Reference: vello_shaders/shader/binning.wgsl:55-203, tile_alloc.wgsl:35-123
// (vello/src/render.rs:405-436); CPU twins cpu/binning.rs, cpu/tile_alloc.rs.
#include "binning_body.h"

namespace vk {

__global__ void __launch_bounds__(256) k_binning(Config cfg, const DrawMonoid *__restrict__ draw_monoids,
                                                 const PathBbox *__restrict__ path_bbox_buf, const Bbox4 *__restrict__ clip_bbox_buf,
                                                 Bbox4 *__restrict__ intersected_bbox, Bump *bump, uint32_t *__restrict__ bin_data,
                                                 BinHeader *__restrict__ bin_header) {
    binning_workgroup(cfg, blockIdx.x, draw_monoids, path_bbox_buf, clip_bbox_buf, intersected_bbox, bump, bin_data, bin_header);
}

__global__ void __launch_bounds__(256) k_tile_alloc(Config cfg, const uint32_t *__restrict__ scene, const Bbox4 *__restrict__ draw_bboxes,
                                                    Bump *bump, Path *__restrict__ paths, Tile *__restrict__ tiles) {
    tile_alloc_workgroup(cfg, blockIdx.x, scene, draw_bboxes, bump, paths, tiles);
}

// binning and tile_alloc of the same 256 draw objects as ONE workgroup (round 6): tile_alloc reads of binning's results only the
// draw_bboxes of its own 256 objects (tile_alloc.wgsl:47-62), which this workgroup has just written -- no other workgroup is waited
// for, the two stages' bump allocations are independent.  One launch boundary less on a large scene's critical path (a small one's
// front stages share launches in k_front, flatten.hip).
__global__ void __launch_bounds__(256) k_binning_tile_alloc(Config cfg, const DrawMonoid *__restrict__ draw_monoids,
                                                            const PathBbox *__restrict__ path_bbox_buf, const Bbox4 *__restrict__ clip_bbox_buf,
                                                            Bbox4 *intersected_bbox, Bump *bump, uint32_t *__restrict__ bin_data,
                                                            BinHeader *__restrict__ bin_header, const uint32_t *__restrict__ scene,
                                                            Path *__restrict__ paths, Tile *__restrict__ tiles, uint32_t n_binning_wg) {
    // (flatten's overflow, which binning detects -- binning.wgsl:64-75 --, is the same answer in every workgroup: tile_alloc does
    // not run, as in the reference, where it finds STAGE_FLATTEN set.  A bin_data overflow is known only to the workgroup it
    // happens in: the others still allocate their tiles -- the frame fails either way, and bump.tile of a failed frame is then a
    // lower bound of the demand instead of zero.)
    if (bump->lines > cfg.lines_size) {
        if (blockIdx.x == 0u && threadIdx.x == 0u) atomicOr(&bump->failed, STAGE_FLATTEN);
        return;
    }
    if (blockIdx.x < n_binning_wg)
        binning_workgroup(cfg, blockIdx.x, draw_monoids, path_bbox_buf, clip_bbox_buf, intersected_bbox, bump, bin_data, bin_header);
    __syncthreads();  // (the workgroup's draw_bboxes are written and visible to it: a barrier is a workgroup-scope fence; binning's LDS is free)
    tile_alloc_workgroup(cfg, blockIdx.x, scene, intersected_bbox, bump, paths, tiles);
}

void launch_binning_tile_alloc(const Frame &f, hipStream_t s) {
    const uint32_t n_b = (f.cfg.layout.n_draw_objects + 255u) / 256u, n_t = (f.cfg.layout.n_paths + 255u) / 256u;
    const uint32_t n_wg = n_b > n_t ? n_b : n_t;
    if (n_wg == 0) return;
    hipLaunchKernelGGL(k_binning_tile_alloc, dim3(n_wg), dim3(256), 0, s, f.cfg, f.draw_monoids, f.path_bboxes, f.clip_bboxes, f.draw_bboxes,
                       f.bump(), f.info_bin_data, f.bin_headers, f.scene, f.paths, f.tiles, n_b);
}

void launch_binning(const Frame &f, hipStream_t s) {
    uint32_t n_wg = (f.cfg.layout.n_draw_objects + 255u) / 256u;
    if (n_wg == 0) return;
    hipLaunchKernelGGL(k_binning, dim3(n_wg), dim3(256), 0, s, f.cfg, f.draw_monoids, f.path_bboxes, f.clip_bboxes, f.draw_bboxes,
                       f.bump(), f.info_bin_data, f.bin_headers);
}

void launch_tile_alloc(const Frame &f, hipStream_t s) {
    uint32_t n_wg = (f.cfg.layout.n_paths + 255u) / 256u;
    if (n_wg == 0) return;
    hipLaunchKernelGGL(k_tile_alloc, dim3(n_wg), dim3(256), 0, s, f.cfg, f.scene, f.draw_bboxes, f.bump(), f.paths, f.tiles);
}

}  // namespace vk
