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
