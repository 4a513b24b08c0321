// draw-object scan (+ draw_leaf) and clip matching.
// Reference: draw_reduce.wgsl:22-55 + draw_leaf.wgsl:52-289 (vello/src/render.rs:333-358) and
// clip_reduce.wgsl:24-67 + clip_leaf.wgsl:80-217 (render.rs:368-393); CPU twins
// vello_shaders/src/cpu/{draw_reduce,draw_leaf,clip_reduce,clip_leaf}.rs.
#include "draw_scan.h"

namespace vk {

__global__ void __launch_bounds__(256) k_draw_scan(Config cfg, const uint32_t *__restrict__ scene, Control *control,
                                                   unsigned long long *state, const PathBbox *__restrict__ path_bbox,
                                                   DrawMonoid *__restrict__ draw_monoid, uint32_t *__restrict__ info,
                                                   Clip *__restrict__ clip_inp) {
    draw_scan_workgroup(cfg, scene, control, state, path_bbox, draw_monoid, info, clip_inp);
}

// Clip matching.  The reference resolves the push/pop stack with a bicyclic-semigroup reduce +
// a 256-wide tree search per partition (clip_leaf.wgsl:37-68).  Clip counts are tiny next to
// everything else in a frame (0 for every BASELINE config), so this round uses the sequential
// stack machine of cpu/clip_leaf.rs:21-72 on ONE wave: 64 lanes gather a chunk of clip inputs and
// their path bboxes into LDS in parallel, lane 0 runs the stack from LDS only (~40 cycles per
// element), then all lanes scatter the clip bboxes and the EndClip draw-monoid patches.
constexpr uint32_t CLIP_CHUNK = 64;
constexpr uint32_t CLIP_LDS_STACK = 1024;

__global__ void __launch_bounds__(64) k_clip(Config cfg, const Clip *__restrict__ clip_inp, const PathBbox *__restrict__ path_bboxes,
                                             DrawMonoid *draw_monoids, Bbox4 *__restrict__ clip_bboxes, uint32_t *clip_stack_spill) {
    __shared__ int32_t sh_path_ix[CLIP_CHUNK];
    __shared__ uint32_t sh_ix[CLIP_CHUNK];
    __shared__ float sh_pb[CLIP_CHUNK][4];
    __shared__ float sh_out_bbox[CLIP_CHUNK][4];
    __shared__ uint32_t sh_out_parent[CLIP_CHUNK];  // END: draw object of the matching BeginClip (~0 = none)
    __shared__ uint32_t sh_out_path[CLIP_CHUNK];
    __shared__ uint32_t st_parent[CLIP_LDS_STACK];
    __shared__ uint32_t st_path[CLIP_LDS_STACK];
    __shared__ float st_bbox[CLIP_LDS_STACK][4];
    __shared__ uint32_t sh_sp;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_clips = cfg.layout.n_clips;
    if (lane == 0) sh_sp = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n_clips; base += CLIP_CHUNK) {
        uint32_t gi = base + lane;
        if (gi < n_clips) {
            Clip c = clip_inp[gi];
            sh_path_ix[lane] = c.path_ix;
            sh_ix[lane] = c.ix;
            if (c.path_ix >= 0) {
                PathBbox pb = path_bboxes[c.path_ix];
                sh_pb[lane][0] = (float)pb.x0; sh_pb[lane][1] = (float)pb.y0;
                sh_pb[lane][2] = (float)pb.x1; sh_pb[lane][3] = (float)pb.y1;
            }
        }
        __syncthreads();
        if (lane == 0) {
            uint32_t sp = sh_sp;
            uint32_t n = minu(CLIP_CHUNK, n_clips - base);
            for (uint32_t i = 0; i < n; i++) {
                int32_t pix = sh_path_ix[i];
                if (pix >= 0) {
                    float b0 = sh_pb[i][0], b1 = sh_pb[i][1], b2 = sh_pb[i][2], b3 = sh_pb[i][3];
                    if (sp > 0u) {
                        float l0, l1, l2, l3;
                        uint32_t t = sp - 1u;
                        if (t < CLIP_LDS_STACK) {
                            l0 = st_bbox[t][0]; l1 = st_bbox[t][1]; l2 = st_bbox[t][2]; l3 = st_bbox[t][3];
                        } else {
                            const uint32_t *g = clip_stack_spill + (size_t)t * 6u;
                            l0 = __uint_as_float(g[2]); l1 = __uint_as_float(g[3]);
                            l2 = __uint_as_float(g[4]); l3 = __uint_as_float(g[5]);
                        }
                        b0 = maxf(b0, l0); b1 = maxf(b1, l1); b2 = minf(b2, l2); b3 = minf(b3, l3);
                    }
                    sh_out_bbox[i][0] = b0; sh_out_bbox[i][1] = b1; sh_out_bbox[i][2] = b2; sh_out_bbox[i][3] = b3;
                    sh_out_parent[i] = ~0u;
                    if (sp < CLIP_LDS_STACK) {
                        st_parent[sp] = sh_ix[i];
                        st_path[sp] = (uint32_t)pix;
                        st_bbox[sp][0] = b0; st_bbox[sp][1] = b1; st_bbox[sp][2] = b2; st_bbox[sp][3] = b3;
                    } else {
                        uint32_t *g = clip_stack_spill + (size_t)sp * 6u;
                        g[0] = sh_ix[i]; g[1] = (uint32_t)pix;
                        g[2] = __float_as_uint(b0); g[3] = __float_as_uint(b1);
                        g[4] = __float_as_uint(b2); g[5] = __float_as_uint(b3);
                    }
                    sp++;
                } else {
                    sh_out_parent[i] = ~0u;
                    sh_out_bbox[i][0] = -1e9f; sh_out_bbox[i][1] = -1e9f; sh_out_bbox[i][2] = 1e9f; sh_out_bbox[i][3] = 1e9f;
                    if (sp == 0u) continue;  // unbalanced; resolve() guarantees this cannot happen
                    sp--;
                    uint32_t tos_parent, tos_path;
                    if (sp < CLIP_LDS_STACK) {
                        tos_parent = st_parent[sp];
                        tos_path = st_path[sp];
                    } else {
                        const uint32_t *g = clip_stack_spill + (size_t)sp * 6u;
                        tos_parent = g[0];
                        tos_path = g[1];
                    }
                    if (sp > 0u) {
                        uint32_t t = sp - 1u;
                        if (t < CLIP_LDS_STACK) {
                            sh_out_bbox[i][0] = st_bbox[t][0]; sh_out_bbox[i][1] = st_bbox[t][1];
                            sh_out_bbox[i][2] = st_bbox[t][2]; sh_out_bbox[i][3] = st_bbox[t][3];
                        } else {
                            const uint32_t *g = clip_stack_spill + (size_t)t * 6u;
                            sh_out_bbox[i][0] = __uint_as_float(g[2]); sh_out_bbox[i][1] = __uint_as_float(g[3]);
                            sh_out_bbox[i][2] = __uint_as_float(g[4]); sh_out_bbox[i][3] = __uint_as_float(g[5]);
                        }
                    }
                    sh_out_parent[i] = tos_parent;
                    sh_out_path[i] = tos_path;
                }
            }
            sh_sp = sp;
        }
        __syncthreads();
        if (gi < n_clips) {
            Bbox4 b = {sh_out_bbox[lane][0], sh_out_bbox[lane][1], sh_out_bbox[lane][2], sh_out_bbox[lane][3]};
            clip_bboxes[gi] = b;
            uint32_t parent = sh_out_parent[lane];
            if (sh_path_ix[lane] < 0 && parent != ~0u) {
                uint32_t ix = sh_ix[lane];
                draw_monoids[ix].path_ix = sh_out_path[lane];
                draw_monoids[ix].scene_offset = draw_monoids[parent].scene_offset;
                draw_monoids[ix].info_offset = draw_monoids[parent].info_offset;
            }
        }
        __syncthreads();
    }
}

void launch_draw_scan(const Frame &f, hipStream_t s) {
    uint32_t n_parts = (f.cfg.layout.n_draw_objects + DRAW_PART - 1u) / DRAW_PART;
    if (n_parts == 0) return;
    hipLaunchKernelGGL(k_draw_scan, dim3(n_parts), dim3(256), 0, s, f.cfg, f.scene, f.control, f.draw_state, f.path_bboxes,
                       f.draw_monoids, f.info_bin_data, f.clip_inp);
}

void launch_clip_sequential(const Frame &f, hipStream_t s) {
    if (f.cfg.layout.n_clips == 0) return;
    hipLaunchKernelGGL(k_clip, dim3(1), dim3(64), 0, s, f.cfg, f.clip_inp, f.path_bboxes, f.draw_monoids, f.clip_bboxes,
                       f.clip_stack);
}

}  // namespace vk
