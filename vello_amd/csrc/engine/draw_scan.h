// draw-object scan (+ draw_leaf): the body of the draw stage, shared by its two launch shapes.
// Reference: draw_reduce.wgsl:22-55 + draw_leaf.wgsl:52-289 (vello/src/render.rs:333-358); CPU twins
// vello_shaders/src/cpu/{draw_reduce,draw_leaf}.rs.
#pragma once
#include "lookback.h"

namespace vk {

namespace draw_detail {

__device__ __forceinline__ Xform xf_inverse(const Xform &t) {  // shared/transform.wgsl:13-18
    Xform r;
    float inv_det = 1.0f / (t.m0 * t.m3 - t.m1 * t.m2);
    r.m0 = inv_det * t.m3;
    r.m1 = inv_det * -t.m1;
    r.m2 = inv_det * -t.m2;
    r.m3 = inv_det * t.m0;
    float tx = -t.t0, ty = -t.t1;
    r.t0 = r.m0 * tx + r.m2 * ty;
    r.t1 = r.m1 * tx + r.m3 * ty;
    return r;
}
__device__ __forceinline__ Xform xf_mul(const Xform &a, const Xform &b) {  // shared/transform.wgsl:20-25
    Xform r;
    r.m0 = a.m0 * b.m0 + a.m2 * b.m1;
    r.m1 = a.m1 * b.m0 + a.m3 * b.m1;
    r.m2 = a.m0 * b.m2 + a.m2 * b.m3;
    r.m3 = a.m1 * b.m2 + a.m3 * b.m3;
    r.t0 = a.m0 * b.t0 + a.m2 * b.t1 + a.t0;
    r.t1 = a.m1 * b.t0 + a.m3 * b.t1 + a.t1;
    return r;
}
__device__ __forceinline__ vec2 xf_apply_plain(const Xform &t, vec2 p) {  // shared/transform.wgsl:9-11
    return v2(t.m0 * p.x + t.m2 * p.y + t.t0, t.m1 * p.x + t.m3 * p.y + t.t1);
}
__device__ __forceinline__ Xform from_poly2(vec2 p0, vec2 p1) {
    return Xform{p1.y - p0.y, p0.x - p1.x, p1.x - p0.x, p1.y - p0.y, p0.x, p0.y};
}
__device__ __forceinline__ Xform two_point_to_unit_line(vec2 p0, vec2 p1) {
    Xform tmp1 = from_poly2(p0, p1);
    Xform inv = xf_inverse(tmp1);
    Xform tmp2 = from_poly2(v2(0.0f, 0.0f), v2(1.0f, 0.0f));
    return xf_mul(tmp2, inv);
}
__device__ __forceinline__ void write_xform(uint32_t *info, const Xform &x) {
    info[0] = __float_as_uint(x.m0); info[1] = __float_as_uint(x.m1); info[2] = __float_as_uint(x.m2);
    info[3] = __float_as_uint(x.m3); info[4] = __float_as_uint(x.t0); info[5] = __float_as_uint(x.t1);
}

}  // namespace draw_detail

// The work of one workgroup of the draw stage: exclusive scan of the draw monoid (decoupled look-back, 256 objects per
// workgroup, partitions handed out by ticket) followed by the draw_leaf body for the object this thread owns.  Reads the
// scene, and of the frame's buffers only the draw flags / transform index of each path (k_pathtag_scan writes them at the
// PATH markers): the workgroups ride in k_flatten_light's launch (flatten.hip) when the two stages run in one go, and are
// k_draw_scan (draw.hip) when the draw stage is run on its own.
__device__ __forceinline__ void draw_scan_workgroup(const Config &cfg, const uint32_t *__restrict__ scene, Control *control,
                                                    unsigned long long *state, const PathBbox *__restrict__ path_bbox,
                                                    DrawMonoid *__restrict__ draw_monoid, uint32_t *__restrict__ info,
                                                    Clip *__restrict__ clip_inp) {
    using namespace draw_detail;
    __shared__ uint32_t sh_part;
    __shared__ uint32_t sh_wave[4][4];
    __shared__ uint32_t sh_excl[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) sh_part = atomicAdd(&control->ticket_draw, 1u);
    __syncthreads();
    const uint32_t part = sh_part;
    const uint32_t ix = part * DRAW_PART + (uint32_t)tid;
    const uint32_t n_draw = cfg.layout.n_draw_objects;
    uint32_t tag_word = ix < n_draw ? scene[cfg.layout.draw_tag_base + ix] : DRAWTAG_NOP;
    // map_draw_tag, shared/drawtag.wgsl:47-54
    uint32_t mv[4] = {tag_word != DRAWTAG_NOP ? 1u : 0u, tag_word & 1u, (tag_word >> 2) & 0x07u, (tag_word >> 6) & 0x0fu};
    uint32_t inc[4];
#pragma unroll
    for (int f = 0; f < 4; f++) inc[f] = wave_incl_scan_u32(mv[f], lane);
    if (lane == 63) {
#pragma unroll
        for (int f = 0; f < 4; f++) sh_wave[w][f] = inc[f];
    }
    __syncthreads();
    uint32_t wave_excl[4], block_agg[4];
#pragma unroll
    for (int f = 0; f < 4; f++) {
        uint32_t s0 = sh_wave[0][f], s1 = sh_wave[1][f], s2 = sh_wave[2][f], s3 = sh_wave[3][f];
        wave_excl[f] = (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u);
        block_agg[f] = s0 + s1 + s2 + s3;
    }
    if (w == 0) {
        uint32_t excl[4];
        decoupled_lookback<4>(state, part, block_agg, excl, &control->bump.failed);
        if (lane == 0) {
#pragma unroll
            for (int f = 0; f < 4; f++) sh_excl[f] = excl[f];
        }
    }
    __syncthreads();
    DrawMonoid m;
    m.path_ix = sh_excl[0] + wave_excl[0] + (inc[0] - mv[0]);
    m.clip_ix = sh_excl[1] + wave_excl[1] + (inc[1] - mv[1]);
    m.scene_offset = sh_excl[2] + wave_excl[2] + (inc[2] - mv[2]);
    m.info_offset = sh_excl[3] + wave_excl[3] + (inc[3] - mv[3]);
    if (ix >= n_draw) return;
    draw_monoid[ix] = m;

    // draw_leaf.wgsl:105-288
    const uint32_t dd = cfg.layout.draw_data_base + m.scene_offset;
    const uint32_t di = m.info_offset;
    if (tag_word == DRAWTAG_FILL_COLOR || tag_word == DRAWTAG_FILL_LIN_GRADIENT || tag_word == DRAWTAG_FILL_RAD_GRADIENT ||
        tag_word == DRAWTAG_FILL_SWEEP_GRADIENT || tag_word == DRAWTAG_FILL_IMAGE || tag_word == DRAWTAG_BEGIN_CLIP ||
        tag_word == DRAWTAG_BLURRED_ROUNDED_RECT) {
        PathBbox bbox = path_bbox[m.path_ix];
        uint32_t draw_flags = bbox.draw_flags;
        if (tag_word == DRAWTAG_FILL_COLOR || tag_word == DRAWTAG_BEGIN_CLIP) {
            info[di] = draw_flags;
        } else {
            Xform transform = read_transform(scene, cfg.layout.transform_base, bbox.trans_ix);
            info[di] = draw_flags;
            if (tag_word == DRAWTAG_FILL_LIN_GRADIENT) {
                vec2 p0 = v2(__uint_as_float(scene[dd + 1]), __uint_as_float(scene[dd + 2]));
                vec2 p1 = v2(__uint_as_float(scene[dd + 3]), __uint_as_float(scene[dd + 4]));
                p0 = xf_apply_plain(transform, p0);
                p1 = xf_apply_plain(transform, p1);
                vec2 dxy = p1 - p0;
                float scale = 1.0f / dot(dxy, dxy);
                vec2 line_xy = dxy * scale;
                float line_c = -dot(p0, line_xy);
                info[di + 1] = __float_as_uint(line_xy.x);
                info[di + 2] = __float_as_uint(line_xy.y);
                info[di + 3] = __float_as_uint(line_c);
            } else if (tag_word == DRAWTAG_FILL_RAD_GRADIENT) {
                const float GRADIENT_EPSILON = 1.0f / (float)(1 << 12);
                vec2 p0 = v2(__uint_as_float(scene[dd + 1]), __uint_as_float(scene[dd + 2]));
                vec2 p1 = v2(__uint_as_float(scene[dd + 3]), __uint_as_float(scene[dd + 4]));
                float r0 = __uint_as_float(scene[dd + 5]);
                float r1 = __uint_as_float(scene[dd + 6]);
                Xform user_to_gradient = xf_inverse(transform);
                Xform xform;
                float focal_x = 0.0f, radius;
                uint32_t kind, flags = 0u;
                if (fabsf(r0 - r1) < GRADIENT_EPSILON) {
                    kind = RAD_GRAD_KIND_STRIP;
                    float scaled = r0 / length(p0 - p1);
                    xform = xf_mul(two_point_to_unit_line(p0, p1), user_to_gradient);
                    radius = scaled * scaled;
                } else {
                    kind = RAD_GRAD_KIND_CONE;
                    if (p0.x == p1.x && p0.y == p1.y) {
                        kind = RAD_GRAD_KIND_CIRCULAR;
                        p0.x += GRADIENT_EPSILON;
                    }
                    if (r1 == 0.0f) {
                        flags |= RAD_GRAD_SWAPPED;
                        vec2 tp = p0; p0 = p1; p1 = tp;
                        float tr = r0; r0 = r1; r1 = tr;
                    }
                    focal_x = r0 / (r0 - r1);
                    vec2 cf = p0 * (1.0f - focal_x) + p1 * focal_x;
                    radius = r1 / length(cf - p1);
                    Xform user_to_unit_line = xf_mul(two_point_to_unit_line(cf, p1), user_to_gradient);
                    Xform sc = Xform{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    if (fabsf(radius - 1.0f) <= GRADIENT_EPSILON) {
                        kind = RAD_GRAD_KIND_FOCAL_ON_CIRCLE;
                        float scale = 0.5f * fabsf(1.0f - focal_x);
                        sc.m0 = scale; sc.m3 = scale;
                    } else {
                        float a = radius * radius - 1.0f;
                        float scale_ratio = fabsf(1.0f - focal_x) / a;
                        sc.m0 = radius * scale_ratio;
                        sc.m3 = sqrtf(fabsf(a)) * scale_ratio;
                    }
                    xform = xf_mul(sc, user_to_unit_line);
                }
                write_xform(info + di + 1, xform);
                info[di + 7] = __float_as_uint(focal_x);
                info[di + 8] = __float_as_uint(radius);
                info[di + 9] = (flags << 3) | kind;
            } else if (tag_word == DRAWTAG_FILL_SWEEP_GRADIENT) {
                vec2 p0 = v2(__uint_as_float(scene[dd + 1]), __uint_as_float(scene[dd + 2]));
                Xform tr = Xform{1.0f, 0.0f, 0.0f, 1.0f, p0.x, p0.y};
                Xform xform = xf_inverse(xf_mul(transform, tr));
                write_xform(info + di + 1, xform);
                info[di + 7] = scene[dd + 3];
                info[di + 8] = scene[dd + 4];
            } else if (tag_word == DRAWTAG_FILL_IMAGE) {
                Xform xform = xf_inverse(transform);
                write_xform(info + di + 1, xform);
                info[di + 7] = scene[dd];
                info[di + 8] = scene[dd + 1];
                info[di + 9] = scene[dd + 2];
            } else {  // DRAWTAG_BLURRED_ROUNDED_RECT
                Xform xform = xf_inverse(transform);
                write_xform(info + di + 1, xform);
                info[di + 7] = scene[dd + 1];
                info[di + 8] = scene[dd + 2];
                info[di + 9] = scene[dd + 3];
                info[di + 10] = scene[dd + 4];
            }
        }
    }
    if (tag_word == DRAWTAG_BEGIN_CLIP || tag_word == DRAWTAG_END_CLIP) {
        uint32_t path_ix = ~ix;
        if (tag_word == DRAWTAG_BEGIN_CLIP) path_ix = m.path_ix;
        Clip c;
        c.ix = ix;
        c.path_ix = (int32_t)path_ix;
        clip_inp[m.clip_ix] = c;
    }
}

}  // namespace vk
