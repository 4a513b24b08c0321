// Clip matching in parallel: which BeginClip every EndClip closes, and for every clip the intersection of the path
// bounding boxes of all layers open at that point.
// Reference: clip_reduce.wgsl:24-67 + clip_leaf.wgsl:80-217 (vello/src/render.rs:368-393), ClipBic::combine
// (vello_encoding/src/clip.rs:69-72); results are defined by the sequential stack of cpu/clip_leaf.rs:21-72, which
// the oracle restates (oracle/vo_front.c: vo_stage_clip).  min / max are exact, so the order of the intersections
// does not show in the result.
//
// The reference holds a 256-deep stack window and the reduced values of at most 256 partitions in one workgroup
// (clip_leaf.wgsl:87-112: deeper stacks and clip 65 537 onward are TODOs there).  Neither limit exists here:
//   k_clip_parts<REDUCE>  a workgroup per 256 clips, a thread per clip.  Local depth D = prefix sum of +1 / -1; the
//                         parent of a clip is (the last earlier position whose depth lies below a threshold) + 1,
//                         found in a min-tree over D in LDS; a clip whose parent is not in the partition knows how far
//                         below the top of the INCOMING stack it sits (its "exit depth").  Pointer jumping over the
//                         parent links (8 rounds in LDS) intersects the boxes of the local ancestors.  Out: the
//                         partition's Bic (a = pops that reach below the partition, b = pushes left open) and its b
//                         open pushes in stack order with their local intersections.
//   k_clip_stack          ONE workgroup over the partitions (<= 2048 = 524 288 clips): scan of the Bics -> stack height
//                         H at every partition start and base = height after the partition's pops; the entry below
//                         base[k] was pushed by the last earlier partition with a smaller base (min-tree search again),
//                         which makes the partitions a tree; pointer jumping over it gives G[k] = the full
//                         intersection of everything below partition k's own pushes.
//   k_clip_parts<LEAF>    the same local pass, then each partition fetches the <= a + 1 entries of the incoming stack
//                         that its clips can see (entry e belongs to the last earlier partition with base <= e) and
//                         finishes: BeginClip = local intersection ∩ window box; EndClip = the box of what is on top
//                         after the pop, and the DrawMonoid patch of clip_leaf.wgsl:195-204.
// Scenes of <= 256 clips run the LEAF launch alone (empty incoming stack).  Beyond 2048 partitions the one-wave stack
// machine of draw.hip (k_clip) takes over; VELLO_HIP_DEBUG_SEQ_CLIP forces it for the tests.
// An EndClip on an empty stack is skipped, as in k_clip (resolve.rs:127-129 closes open layers, nothing removes extra pops).
#include "engine.h"

namespace vk {

namespace {

constexpr int32_t CLIP_INF = 0x7fffffff;
constexpr float CLIP_BIG = 1e9f;  // clip_leaf.wgsl:128

__device__ __forceinline__ uint32_t tree_off(uint32_t n_pad, uint32_t level) { return 2u * n_pad - ((2u * n_pad) >> level); }

// Min-tree over n_pad (a power of two) values: level 0 = the values, level l = minima of aligned blocks of 2^l.
// Returns the LAST index q < i whose value is < v, or -1.
template <typename P>
__device__ __forceinline__ int32_t last_below(P tree, uint32_t n_pad, uint32_t i, int32_t v) {
    uint32_t level = 0;
    for (uint32_t pos = i; pos != 0u; pos >>= 1, level++) {
        if ((pos & 1u) == 0u) continue;
        if (tree[tree_off(n_pad, level) + pos - 1u] < v) {  // [0, i) = the blocks (i >> l) - 1 of the set bits l of i, nearest first
            uint32_t idx = pos - 1u;
            while (level > 0u) {
                level--;
                idx = idx * 2u + 1u;
                if (!(tree[tree_off(n_pad, level) + idx] < v)) idx--;
            }
            return (int32_t)idx;
        }
    }
    return -1;
}

__device__ __forceinline__ Bbox4 bbox_intersect(Bbox4 a, Bbox4 b) {
    return Bbox4{maxf(a.x0, b.x0), maxf(a.y0, b.y0), minf(a.x1, b.x1), minf(a.y1, b.y1)};
}
__device__ __forceinline__ Bbox4 big_bbox() { return Bbox4{-CLIP_BIG, -CLIP_BIG, CLIP_BIG, CLIP_BIG}; }

}  // namespace

struct ClipBic { uint32_t a, b; };
struct ClipScratch {
    ClipEl *els;       // [parts][256]: a partition's open pushes, bottom first
    ClipBic *bic;      // [parts]
    uint32_t *height;  // [parts]: stack height when the partition starts
    int32_t *tree;     // min-tree over base[] (2 * n_pad entries)
    Bbox4 *below;      // [parts]: G
};

__device__ __forceinline__ ClipScratch clip_scratch(uint32_t *p, uint32_t n_clips) {
    const uint32_t parts = (n_clips + CLIP_PART - 1u) / CLIP_PART;
    ClipScratch s;
    s.els = (ClipEl *)p;
    p += (size_t)parts * CLIP_PART * (sizeof(ClipEl) / 4u);
    s.below = (Bbox4 *)p;  // 16-byte aligned: 256 * 20 bytes per partition
    p += (size_t)parts * 4u;
    s.bic = (ClipBic *)p;
    p += (size_t)parts * 2u;
    s.height = p;
    p += parts;
    s.tree = (int32_t *)p;  // 2 * clip_parts_pad(parts) entries
    return s;
}

template <bool LEAF>
__global__ void __launch_bounds__(CLIP_PART) k_clip_parts(Config cfg, const Clip *__restrict__ clip_inp, const PathBbox *__restrict__ path_bboxes,
                                                           DrawMonoid *draw_monoids, Bbox4 *__restrict__ clip_bboxes, uint32_t *scratch_words,
                                                           uint32_t multi_part) {
    __shared__ int32_t sh_tree[2 * CLIP_PART];
    __shared__ uint32_t sh_scan[4];
    __shared__ int32_t sh_link0[CLIP_PART];
    __shared__ int32_t sh_link[CLIP_PART];
    __shared__ Bbox4 sh_bbox[CLIP_PART];
    __shared__ uint32_t sh_matched[CLIP_PART];
    __shared__ uint32_t sh_win_clip[CLIP_PART + 1];
    __shared__ Bbox4 sh_win_bbox[CLIP_PART + 1];
    const uint32_t t = threadIdx.x;
    const uint32_t part = blockIdx.x;
    const uint32_t n_clips = cfg.layout.n_clips;
    const uint32_t gi = part * CLIP_PART + t;
    const ClipScratch sc = clip_scratch(scratch_words, n_clips);

    Clip inp = {0u, 0};
    const bool valid = gi < n_clips;
    if (valid) inp = clip_inp[gi];
    const bool is_push = valid && inp.path_ix >= 0;
    const bool is_pop = valid && inp.path_ix < 0;
    Bbox4 bbox = big_bbox();
    if (is_push) {
        PathBbox pb = path_bboxes[inp.path_ix];
        bbox = Bbox4{(float)pb.x0, (float)pb.y0, (float)pb.x1, (float)pb.y1};
    }
    // local depth after every clip
    uint32_t total_u;
    const int32_t depth = (int32_t)block256_incl_scan_u32(is_push ? 1u : (is_pop ? 0xffffffffu : 0u), sh_scan, &total_u);
    sh_tree[t] = depth;
    sh_matched[t] = 0u;
    for (uint32_t level = 1; level <= 8u; level++) {
        __syncthreads();
        if (t < (CLIP_PART >> level)) {
            uint32_t src = tree_off(CLIP_PART, level - 1u) + 2u * t;
            sh_tree[tree_off(CLIP_PART, level) + t] = mini(sh_tree[src], sh_tree[src + 1u]);
        }
    }
    __syncthreads();
    const int32_t lowest = mini(0, sh_tree[tree_off(CLIP_PART, 8u)]);  // -a
    // Parent link.  A BeginClip that ends at depth d hangs under the push that made depth d - 1; an EndClip that ends at
    // depth d closes the push that made depth d + 1.  Either is the position after the last one whose depth is below
    // theta = d - 1 / d + 1.  None in the partition and theta <= 0: the parent is entry -theta of the incoming stack,
    // counted from its top; stored as link = -1 - (-theta).
    const int32_t theta = is_push ? depth - 1 : depth + 1;
    int32_t link = -1;
    if (valid) {
        int32_t q = last_below(sh_tree, CLIP_PART, t, theta);
        if (q >= 0) link = q + 1;
        else if (theta > 0) link = 0;  // the depth before the partition's first clip is 0
        else link = theta - 1;
    }
    const int32_t link0 = link;
    sh_link0[t] = link0;
    // intersection of the boxes of the local ancestors (EndClips carry the neutral box: theirs is their BeginClip's chain)
    for (uint32_t round = 0; round < 8u; round++) {
        sh_bbox[t] = bbox;
        sh_link[t] = link;
        __syncthreads();
        if (link >= 0) {
            bbox = bbox_intersect(sh_bbox[link], bbox);
            link = sh_link[link];
        }
        __syncthreads();
    }
    const uint32_t exit_depth = (uint32_t)(-1 - link);  // where this clip's chain leaves the partition

    if (!LEAF) {
        // pushes that no EndClip of the partition closes are what it leaves on the stack
        if (is_pop && link0 >= 0) sh_matched[link0] = 1u;
        __syncthreads();
        if (is_push && sh_matched[t] == 0u) {
            uint32_t j = (uint32_t)(depth - lowest - 1);
            ClipEl el = {gi, bbox.x0, bbox.y0, bbox.x1, bbox.y1};
            sc.els[(size_t)part * CLIP_PART + j] = el;
        }
        if (t == 0u) sc.bic[part] = ClipBic{(uint32_t)(-lowest), (uint32_t)((int32_t)total_u - lowest)};
        return;
    }

    // the entries of the incoming stack this partition can see: 0 (its top) ... a
    uint32_t win_n = 0u;
    if (multi_part) {
        const uint32_t height = sc.height[part];
        const uint32_t n_pad = clip_parts_pad((n_clips + CLIP_PART - 1u) / CLIP_PART);
        win_n = minu(height, (uint32_t)(-lowest) + 1u);
        for (uint32_t x = t; x < win_n; x += CLIP_PART) {
            uint32_t e = height - 1u - x;
            int32_t owner = last_below(sc.tree, n_pad, part, (int32_t)e + 1);  // the last earlier partition with base <= e pushed it
            sh_win_clip[x] = ~0u;
            sh_win_bbox[x] = big_bbox();
            if (owner < 0) continue;  // cannot happen: e < height
            ClipEl el = sc.els[(size_t)owner * CLIP_PART + (e - (uint32_t)sc.tree[owner])];
            sh_win_clip[x] = el.clip_ix;
            sh_win_bbox[x] = bbox_intersect(Bbox4{el.x0, el.y0, el.x1, el.y1}, sc.below[owner]);
        }
    }
    __syncthreads();
    if (exit_depth < win_n) bbox = bbox_intersect(sh_win_bbox[exit_depth], bbox);
    sh_bbox[t] = bbox;  // now: everything open at this clip, itself included
    __syncthreads();
    if (is_push) clip_bboxes[gi] = bbox;
    if (is_pop) {
        uint32_t parent = ~0u;
        Bbox4 out = big_bbox();
        if (link0 >= 0) {
            parent = part * CLIP_PART + (uint32_t)link0;
            int32_t grand = sh_link0[link0];
            if (grand >= 0) out = sh_bbox[grand];
            else if ((uint32_t)(-1 - grand) < win_n) out = sh_win_bbox[-1 - grand];
        } else {
            uint32_t x = (uint32_t)(-1 - link0);
            if (x < win_n) parent = sh_win_clip[x];
            if (x + 1u < win_n) out = sh_win_bbox[x + 1u];
        }
        clip_bboxes[gi] = out;
        if (parent != ~0u) {  // clip_leaf.wgsl:195-204: the EndClip draws with its BeginClip's path, draw data and info
            Clip pc = clip_inp[parent];
            draw_monoids[inp.ix].path_ix = (uint32_t)pc.path_ix;
            draw_monoids[inp.ix].scene_offset = draw_monoids[pc.ix].scene_offset;
            draw_monoids[inp.ix].info_offset = draw_monoids[pc.ix].info_offset;
        }
    }
}

constexpr uint32_t CLIP_STACK_THREADS = 1024;
constexpr uint32_t CLIP_STACK_PER_THREAD = CLIP_MAX_PARTS / CLIP_STACK_THREADS;

__device__ __forceinline__ ClipBic bic_combine(ClipBic x, ClipBic y) {  // shared/clip.wgsl:9-12
    uint32_t m = minu(x.b, y.a);
    return ClipBic{x.a + y.a - m, x.b + y.b - m};
}

__global__ void __launch_bounds__(CLIP_STACK_THREADS) k_clip_stack(Config cfg, uint32_t *scratch_words) {
    __shared__ int32_t sh_tree[2 * CLIP_MAX_PARTS];
    __shared__ int32_t sh_link[CLIP_MAX_PARTS];
    __shared__ Bbox4 sh_box[CLIP_MAX_PARTS];  // first the Bic scan's buffer, then the boxes
    ClipBic *sh_bic = (ClipBic *)sh_box;
    const uint32_t t = threadIdx.x;
    const uint32_t n_clips = cfg.layout.n_clips;
    const uint32_t parts = (n_clips + CLIP_PART - 1u) / CLIP_PART;
    const uint32_t n_pad = clip_parts_pad(parts);
    const ClipScratch sc = clip_scratch(scratch_words, n_clips);

    // inclusive scan of the partitions' Bics (Hillis-Steele; bic_combine is associative, earlier operand first)
    ClipBic own[CLIP_STACK_PER_THREAD], v[CLIP_STACK_PER_THREAD];
    for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
        uint32_t k = t + r * CLIP_STACK_THREADS;
        own[r] = k < parts ? sc.bic[k] : ClipBic{0u, 0u};
        v[r] = own[r];
        if (k < n_pad) sh_bic[k] = v[r];
    }
    for (uint32_t d = 1; d < n_pad; d <<= 1) {
        __syncthreads();
        ClipBic o[CLIP_STACK_PER_THREAD];
        for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
            uint32_t k = t + r * CLIP_STACK_THREADS;
            if (k < n_pad && k >= d) o[r] = sh_bic[k - d];
        }
        __syncthreads();
        for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
            uint32_t k = t + r * CLIP_STACK_THREADS;
            if (k < n_pad && k >= d) {
                v[r] = bic_combine(o[r], v[r]);
                sh_bic[k] = v[r];
            }
        }
    }
    __syncthreads();
    // height at the partition's start, base = height once its pops are done (pops on an empty stack are skipped)
    uint32_t base[CLIP_STACK_PER_THREAD];
    for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
        uint32_t k = t + r * CLIP_STACK_THREADS;
        if (k >= n_pad) continue;
        uint32_t height = k > 0u ? sh_bic[k - 1u].b : 0u;
        base[r] = height > own[r].a ? height - own[r].a : 0u;
        sh_tree[k] = k < parts ? (int32_t)base[r] : CLIP_INF;
        if (k < parts) sc.height[k] = height;
    }
    for (uint32_t level = 1; (n_pad >> level) != 0u; level++) {
        __syncthreads();
        for (uint32_t i = t; i < (n_pad >> level); i += CLIP_STACK_THREADS) {
            uint32_t src = tree_off(n_pad, level - 1u) + 2u * i;
            sh_tree[tree_off(n_pad, level) + i] = mini(sh_tree[src], sh_tree[src + 1u]);
        }
    }
    __syncthreads();  // also: the scan buffer is free from here on
    for (uint32_t i = t; i < 2u * n_pad; i += CLIP_STACK_THREADS) sc.tree[i] = sh_tree[i];
    // the entry under partition k's pushes (index base - 1) was pushed by the last earlier partition with a smaller base
    int32_t link[CLIP_STACK_PER_THREAD];
    Bbox4 box[CLIP_STACK_PER_THREAD];
    for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
        uint32_t k = t + r * CLIP_STACK_THREADS;
        link[r] = -1;
        box[r] = big_bbox();
        if (k >= parts) continue;
        link[r] = last_below(sh_tree, n_pad, k, (int32_t)base[r]);
        if (link[r] >= 0) {
            ClipEl el = sc.els[(size_t)link[r] * CLIP_PART + (base[r] - 1u - (uint32_t)sh_tree[link[r]])];
            box[r] = Bbox4{el.x0, el.y0, el.x1, el.y1};
        }
    }
    for (uint32_t d = 1; d < n_pad; d <<= 1) {
        for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
            uint32_t k = t + r * CLIP_STACK_THREADS;
            if (k < n_pad) {
                sh_box[k] = box[r];
                sh_link[k] = link[r];
            }
        }
        __syncthreads();
        for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
            if (link[r] >= 0) {
                box[r] = bbox_intersect(sh_box[link[r]], box[r]);
                link[r] = sh_link[link[r]];
            }
        }
        __syncthreads();
    }
    for (uint32_t r = 0; r < CLIP_STACK_PER_THREAD; r++) {
        uint32_t k = t + r * CLIP_STACK_THREADS;
        if (k < parts) sc.below[k] = box[r];
    }
}

void launch_clip(const Frame &f, hipStream_t s) {
    const uint32_t n_clips = f.cfg.layout.n_clips;
    if (n_clips == 0) return;  // render.rs:368,379: clip dispatches are skipped when there are no clips
    const uint32_t parts = (n_clips + CLIP_PART - 1u) / CLIP_PART;
    if (parts > CLIP_MAX_PARTS || f.sequential_clip) {
        launch_clip_sequential(f, s);
        return;
    }
    if (parts > 1u) {
        hipLaunchKernelGGL(k_clip_parts<false>, dim3(parts), dim3(CLIP_PART), 0, s, f.cfg, f.clip_inp, f.path_bboxes, f.draw_monoids,
                           f.clip_bboxes, f.clip_stack, 1u);
        hipLaunchKernelGGL(k_clip_stack, dim3(1), dim3(CLIP_STACK_THREADS), 0, s, f.cfg, f.clip_stack);
    }
    hipLaunchKernelGGL(k_clip_parts<true>, dim3(parts), dim3(CLIP_PART), 0, s, f.cfg, f.clip_inp, f.path_bboxes, f.draw_monoids,
                       f.clip_bboxes, f.clip_stack, parts > 1u ? 1u : 0u);
}

}  // namespace vk
