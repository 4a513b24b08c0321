// pathtag scan + bbox_clear, one launch.
// Reference: pathtag_reduce / pathtag_reduce2 / pathtag_scan1 / pathtag_scan_{small,large} +
// bbox_clear (vello/src/render.rs:250-308, vello_shaders/shader/pathtag_scan.wgsl:28-76,
// shared/pathtag.wgsl:58-71, bbox_clear.wgsl:13-23).
// gfx950 design: 256 threads x 4 tag words per workgroup, tags read once (16 B per lane,
// coalesced), SWAR popcount reduction per word, wave64 shuffle scan + one LDS hop, then a
// single-pass decoupled look-back across workgroups (lookback.h).
#include "scan_body.h"

namespace vk {

__global__ void __launch_bounds__(256) k_pathtag_scan(Config cfg, uint32_t n_tag_words, uint32_t n_scene_words, const uint32_t *__restrict__ scene,
                                                      Control *control, unsigned long long *state,
                                                      TagMonoid *__restrict__ tag_monoids, PathBbox *__restrict__ path_bboxes) {
    pathtag_scan_workgroup(cfg, blockIdx.x, gridDim.x, n_tag_words, n_scene_words, scene, control, state, tag_monoids, path_bboxes);
}

void launch_pathtag_scan(const Frame &f, hipStream_t s) {
    uint32_t n_parts = (f.n_tag_words + PATHTAG_PART_WORDS - 1u) / PATHTAG_PART_WORDS;
    if (n_parts == 0) n_parts = 1;
    hipLaunchKernelGGL(k_pathtag_scan, dim3(n_parts), dim3(256), 0, s, f.cfg, f.n_tag_words, f.n_scene_words, f.scene, f.control,
                       f.pathtag_state, f.tag_monoids, f.path_bboxes);
}

}  // namespace vk
