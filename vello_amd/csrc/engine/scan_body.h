// The body of the pathtag scan (scan.hip has the head comment and the launch): a header because k_front (flatten.hip) runs the same
// workgroups as the first stage of its one launch.
#pragma once
#include "lookback.h"

namespace vk {

__device__ __forceinline__ TagMonoid reduce_tag(uint32_t tag_word) {
    TagMonoid c;
    uint32_t point_count = tag_word & 0x3030303u;
    c.pathseg_ix = __popc((point_count * 7u) & 0x4040404u);
    c.trans_ix = __popc(tag_word & (PATH_TAG_TRANSFORM * 0x1010101u));
    uint32_t n_points = point_count + ((tag_word >> 2) & 0x1010101u);
    uint32_t a = n_points + (n_points & (((tag_word >> 3) & 0x1010101u) * 15u));
    a += a >> 8;
    a += a >> 16;
    c.pathseg_offset = a & 0xffu;
    c.path_ix = __popc(tag_word & (PATH_TAG_PATH * 0x1010101u));
    c.style_ix = __popc(tag_word & (PATH_TAG_STYLE * 0x1010101u)) * STYLE_SIZE_IN_WORDS;
    return c;
}

// One partition of the scan (the partition is handed out by ticket; `block` of `n_blocks` only strides bbox_clear).  A workgroup of
// k_pathtag_scan, or one turn of a workgroup of k_front (flatten.hip: the stages up to tile_alloc as one launch for small scenes).
__device__ __forceinline__ void pathtag_scan_workgroup(const Config &cfg, uint32_t block, uint32_t n_blocks, uint32_t n_tag_words, uint32_t n_scene_words,
                                                       const uint32_t *scene, Control *control, unsigned long long *state, TagMonoid *tag_monoids,
                                                       PathBbox *path_bboxes) {
    __shared__ uint32_t sh_part;
    __shared__ uint32_t sh_wave[4][5];
    __shared__ uint32_t sh_excl[5];
    __shared__ uint32_t sh_markers;  // last partition: PATH markers of the whole stream
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // bbox_clear folded in: grid-stride over paths
    for (uint32_t i = block * 256u + tid; i < cfg.layout.n_paths; i += n_blocks * 256u) {
        path_bboxes[i].x0 = 0x7fffffff;
        path_bboxes[i].y0 = 0x7fffffff;
        path_bboxes[i].x1 = (int32_t)0x80000000;
        path_bboxes[i].y1 = (int32_t)0x80000000;
        // (draw_flags / trans_ix: written below by the thread that holds the path's PATH marker)
    }

    if (tid == 0) sh_part = atomicAdd(&control->ticket_pathtag, 1u);
    __syncthreads();
    const uint32_t part = sh_part;
    const uint32_t word0 = part * PATHTAG_PART_WORDS + (uint32_t)tid * 4u;
    const uint32_t *tags = scene + cfg.layout.path_tag_base;

    uint32_t tw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) tw[k] = (word0 + k < n_tag_words) ? tags[word0 + k] : 0u;

    // thread-local exclusive prefixes
    uint32_t ex[4][5];
    uint32_t acc[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        TagMonoid m = reduce_tag(tw[k]);
        ex[k][0] = acc[0]; ex[k][1] = acc[1]; ex[k][2] = acc[2]; ex[k][3] = acc[3]; ex[k][4] = acc[4];
        acc[0] += m.trans_ix; acc[1] += m.pathseg_ix; acc[2] += m.pathseg_offset; acc[3] += m.style_ix; acc[4] += m.path_ix;
    }
    // wave inclusive scan of thread aggregates, 5 fields
    uint32_t inc[5];
#pragma unroll
    for (int f = 0; f < 5; f++) inc[f] = wave_incl_scan_u32(acc[f], lane);
    if (lane == 63) {
#pragma unroll
        for (int f = 0; f < 5; f++) sh_wave[w][f] = inc[f];
    }
    __syncthreads();
    uint32_t wave_excl[5], block_agg[5];
#pragma unroll
    for (int f = 0; f < 5; f++) {
        uint32_t s0 = sh_wave[0][f], s1 = sh_wave[1][f], s2 = sh_wave[2][f], s3 = sh_wave[3][f];
        wave_excl[f] = (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u);
        block_agg[f] = s0 + s1 + s2 + s3;
    }
    if (w == 0) {
        uint32_t excl[5];
        decoupled_lookback<5>(state, part, block_agg, excl, &control->bump.failed);
        if (lane == 0) {
#pragma unroll
            for (int f = 0; f < 5; f++) sh_excl[f] = excl[f];
            // The last partition holds the totals of the whole tag stream: what flatten is going to index with.
            if (part == n_blocks - 1u) {
                const Layout &L = cfg.layout;
                const uint32_t n_trans = excl[0] + block_agg[0], pathseg_words = excl[2] + block_agg[2];
                const uint32_t style_words = excl[3] + block_agg[3];
                // (PATH markers beyond n_paths are legal -- resolve.rs:127-129 appends one per unclosed layer without
                // counting it -- and flatten guards its per-path stores instead)
                const bool ok = (uint64_t)n_trans * 6u <= (uint64_t)(L.style_base - L.transform_base) &&
                                pathseg_words <= L.draw_tag_base - L.path_data_base &&
                                (uint64_t)L.style_base + style_words <= n_scene_words;
                if (!ok) atomicOr(&control->bump.failed, FAILED_SCENE);
                sh_markers = excl[4] + block_agg[4];
            }
        }
    }
    __syncthreads();
    uint32_t base[5];
#pragma unroll
    for (int f = 0; f < 5; f++) base[f] = sh_excl[f] + wave_excl[f] + (inc[f] - acc[f]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (word0 + k < n_tag_words) {
            TagMonoid o;
            o.trans_ix = base[0] + ex[k][0];
            o.pathseg_ix = base[1] + ex[k][1];
            o.pathseg_offset = base[2] + ex[k][2];
            o.style_ix = base[3] + ex[k][3];
            o.path_ix = base[4] + ex[k][4];
            tag_monoids[word0 + k] = o;
            // The reference's flatten stores a path's draw flags and transform index when it meets the PATH marker
            // (flatten.wgsl:813-817); draw_leaf is their only reader.  Stored here, where the marker's monoid is at hand,
            // the draw stage needs nothing of flatten's and its workgroups can share k_flatten_light's launch.
            uint32_t marks = tw[k] & (PATH_TAG_PATH * 0x1010101u);
            while (marks != 0u) {
                const uint32_t shift = ((uint32_t)__ffs((int)marks) - 1u) & ~7u;  // bit offset of the marker's tag byte
                marks &= ~(0xffu << shift);
                const TagMonoid tm = reduce_tag(tw[k] & ((1u << shift) - 1u));  // flatten.wgsl:684-701
                const uint32_t path_ix = o.path_ix + tm.path_ix;
                if (path_ix < cfg.layout.n_paths) {  // a PATH marker per unclosed layer follows the last path (resolve.rs:127-129)
                    const uint32_t style_at = cfg.layout.style_base + (o.style_ix + tm.style_ix - STYLE_SIZE_IN_WORDS);
                    // (a stream that asks for more style words than the scene holds is refused by the last partition)
                    const uint32_t style_flags = style_at < n_scene_words ? scene[style_at] : 0u;
                    path_bboxes[path_ix].draw_flags = (style_flags & STYLE_FLAGS_FILL) == 0u ? 0u : DRAW_INFO_FLAGS_FILL_RULE_BIT;
                    // (likewise for the transforms: an index whose six words -- at the reference's wrapping u32 address; a marker
                    // ahead of the first transform has index -1 -- lie outside the scene buffer is not handed to draw_leaf)
                    const uint32_t trans_ix = o.trans_ix + tm.trans_ix - 1u;
                    const bool trans_ok = (uint64_t)(uint32_t)(cfg.layout.transform_base + trans_ix * 6u) + 6u <= (uint64_t)n_scene_words;
                    path_bboxes[path_ix].trans_ix = trans_ok ? trans_ix : 0u;
                }
            }
        }
    }
    // Paths beyond the stream's last marker (a layout that counts more paths than the tags close) have no writer above:
    // defined values keep draw_leaf's transform read inside the scene.
    if (part == n_blocks - 1u) {
        for (uint32_t i = sh_markers + (uint32_t)tid; i < cfg.layout.n_paths; i += 256u) {
            path_bboxes[i].draw_flags = 0u;
            path_bboxes[i].trans_ix = 0u;
        }
    }
}

}  // namespace vk
