// flatten: path segments -> LineSoup (+ per-path integer bboxes).
// Reference: vello_shaders/shader/flatten.wgsl:328-923 (Euler-spiral flattening of fills, GPU
// stroker with joins/caps); CPU twin vello_shaders/src/cpu/{flatten,euler}.rs.
//
// gfx950 design.  The reference bumps `bump.lines` with one global atomic per emitted line; on
// MI355X same-address device-scope atomics retire at ~12 ns each (MI355X_MICROARCH.md "dequeue"
// row), i.e. ~18 ms for a paris-class frame.  Here each workgroup (256 threads x 4 tags) flattens its
// 1024 tags ONCE into an LDS staging area (one LDS atomic per curve piece reserves its slots), then
// issues ONE atomicAdd for the whole workgroup and copies the staged lines out coalesced.  Pieces
// that do not fit the 3072-line staging area go to the soup directly (one global atomic per piece).
// Line order inside a workgroup follows LDS atomic order, across workgroups global atomic order, as
// in the reference (the line soup is an unordered set, flatten.wgsl:775-798).
#include "engine.h"
#include "draw_scan.h"
#include "scan_body.h"     // (k_front runs the workgroups of the pathtag scan ...
#include "binning_body.h"  //  ... and of binning and tile_alloc)

namespace vk {

namespace {

constexpr float DERIV_THRESH = 1e-6f;
constexpr float DERIV_THRESH_SQUARED = DERIV_THRESH * DERIV_THRESH;
constexpr float DERIV_EPS = 1e-6f;
constexpr float SUBDIV_LIMIT = 1.0f / 65536.0f;
constexpr float K1_THRESH = 1e-3f;
constexpr float DIST_THRESH = 1e-3f;
constexpr float TANGENT_THRESH = 1e-6f;

// Measurement build only (EXTRA=-DVELLO_FLATTEN_PROF, scripts/flatten_prof.py): where the waves of the heavy / stroke / tail
// workgroups spend their cycles.  Lanes of a wave walk different curves, so the timers are the WAVE's: whichever lanes are active
// when the wave passes a mark, its first active lane books the cycles since the wave's previous mark (kept in LDS per wave) to
// the phase, and counts the passage -- the number of times the wave executed that piece of code (the union of its lanes' loops).
// Lane-level event counts (iterations, pieces, lines) go beside them: count / (64 x passages) is the lane use.  Without the
// macro every call is nothing and the product kernels' code is the same with or without these lines.
enum {
    FLP_TAG = 0,     // list entry -> tag monoid -> style / transform / points (flatten_tag's head)
    FLP_SUBDIV,      // flatten_euler: one turn of the subdivision loop up to the accept test (eval_cubic_and_deriv, cubic_from_points_derivs)
    FLP_PIECE,       // ... an accepted range: es_params_from_angles, the per-side integrals, n, the reservation
    FLP_EMIT,        // ... the lines of an accepted range (es_seg_eval_with_offset per line)
    FLP_STRAIGHT,    // ... the straight-segment shortcut
    FLP_JOIN,        // draw_join / draw_cap up to flatten_arc (tangents, atan2)
    FLP_ARC_SETUP,   // flatten_arc: acos, sincos, the reservation
    FLP_ARC_LINES,   // flatten_arc: the rotation loop
    FLP_BBOX_FLUSH,  // wave_bbox_update + the workgroup's flush
    FLP_OTHER,
    FLP_PHASES,
    FLC_ENTRIES = 0, FLC_ITERS, FLC_PIECES, FLC_EULER_LINES, FLC_ARCS, FLC_ARC_LINES, FLC_COUNTS
};
#ifdef VELLO_FLATTEN_PROF
__device__ unsigned long long g_flatten_prof[2 * FLP_PHASES + FLC_COUNTS + 2];  // cycles, passages per phase; lane-level counts; waves, wave cycles
struct FlProfLds {
    unsigned long long prev[4], t0[4];
    uint32_t cyc[4][FLP_PHASES], pass[4][FLP_PHASES], cnt[4][FLC_COUNTS];
};
__device__ __forceinline__ FlProfLds &flp_lds() {
    __shared__ FlProfLds s;
    return s;
}
__device__ __forceinline__ bool flp_leader() {
    const unsigned long long ex = __builtin_amdgcn_read_exec();
    return (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)ex) - 1);
}
__device__ __forceinline__ void flp_start() {
    FlProfLds &s = flp_lds();
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const unsigned long long t = clock64();
    if (l < FLP_PHASES) { s.cyc[w][l] = 0u; s.pass[w][l] = 0u; }
    if (l < FLC_COUNTS) s.cnt[w][l] = 0u;
    if (l == 0u) { s.prev[w] = t; s.t0[w] = t; }
}
__device__ __forceinline__ void flp_mark(int k) {
    const unsigned long long t = clock64();
    if (flp_leader()) {
        FlProfLds &s = flp_lds();
        const uint32_t w = threadIdx.x >> 6;
        s.cyc[w][k] += (uint32_t)(t - s.prev[w]);
        s.pass[w][k] += 1u;
        s.prev[w] = t;
    }
}
__device__ __forceinline__ void flp_count(int k, uint32_t n) { atomicAdd(&flp_lds().cnt[threadIdx.x >> 6][k], n); }
__device__ __forceinline__ void flp_store() {  // whole wave, converged
    FlProfLds &s = flp_lds();
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const unsigned long long t = clock64();
    if (l < FLP_PHASES) {
        atomicAdd(&g_flatten_prof[l], (unsigned long long)s.cyc[w][l]);
        atomicAdd(&g_flatten_prof[FLP_PHASES + l], (unsigned long long)s.pass[w][l]);
    }
    if (l < FLC_COUNTS) atomicAdd(&g_flatten_prof[2 * FLP_PHASES + l], (unsigned long long)s.cnt[w][l]);
    if (l == 0u) {
        atomicAdd(&g_flatten_prof[2 * FLP_PHASES + FLC_COUNTS], 1ull);
        atomicAdd(&g_flatten_prof[2 * FLP_PHASES + FLC_COUNTS + 1], t - s.t0[w]);
    }
}
#else
__device__ __forceinline__ void flp_start() {}
__device__ __forceinline__ void flp_mark(int) {}
__device__ __forceinline__ void flp_count(int, uint32_t) {}
__device__ __forceinline__ void flp_store() {}
#endif


// (out of line ON PURPOSE, round 5: inlined, the fp64 polynomial constants of every copy of these routines are hoisted to the top
// of the kernel as loop invariants -- hundreds of live registers, spilled to scratch and reloaded at every use: 88 VGPRs in the
// cooperative walk, whose every turn then took twice as long.  The price: a call drains every counter at the callee's entry and
// redirects the instruction fetch -- round 4's kernels with nothing changed but these calls are 10-20 % slower (k_flatten_main on
// the road map 64 -> 71 us, r1mix 64 -> 78, mmark-50k 267 -> 321, the tiger 114 -> 124), which is what the road map's flatten pays
// for the cooperative walk's -37 % on mmark-50k and -26 % on the tiger.  Tried and measured, profiles/r05_ab_flatten_coop.txt: the
// walk compiled twice (inline routines for lanes on their own, calls under the cooperative walk) in one kernel -- the inline
// copy's constants spill everywhere, 94 / 100 us -- and the cooperative walk as an out-of-line function of its own with a register
// allocation of its own -- the kernel spills its state around the call, 120 / 116 us.  Larger out-of-line units (a turn's
// cubic_from_points_derivs, a point's es_seg_eval_with_offset, an arc's acos + sincos) take a third of the calls back.)
struct SinCos { float s, c; };
__device__ __attribute__((noinline)) SinCos fl_sincos_call(float x) { SinCos r; sincos_cr(x, r.s, r.c); return r; }
__device__ __forceinline__ void fl_sincos(float x, float &s, float &c) { const SinCos r = fl_sincos_call(x); s = r.s; c = r.c; }
__device__ __forceinline__ float fl_sin(float x) { return fl_sincos_call(x).s; }
__device__ __attribute__((noinline)) float fl_atan2(float y, float x) { return atan2_cr(y, x); }
__device__ __attribute__((noinline)) float fl_asin(float x) { return asin_cr(x); }
__device__ __attribute__((noinline)) float fl_acos(float x) { return acos_cr(x); }
// flatten_arc's theta = max(min_theta, 2 acos(x)) with its sine and cosine: one call for the three
struct ArcMath { float theta, sn, cs; };
__device__ __attribute__((noinline)) ArcMath fl_arc_math(float x, float min_theta) {
    ArcMath r;
    r.theta = maxf(min_theta, 2.0f * acos_cr(x));
    sincos_cr(r.theta, r.sn, r.cs);
    return r;
}
__device__ __forceinline__ ArcMath arc_math_inline(float x, float min_theta) {
    ArcMath r;
    r.theta = maxf(min_theta, 2.0f * acos_cr(x));
    sincos_cr(r.theta, r.sn, r.cs);
    return r;
}
struct CubicParams { float th0, th1, chord_len, err; };
struct EulerParams { float th0, k0, k1, ch; };
struct CubicPoints { vec2 p0, p1, p2, p3; };
struct PointDeriv { vec2 point, deriv; };

// flatten.wgsl:668-672
__device__ __forceinline__ vec2 xf_apply(const Xform &t, vec2 p) {
    float px = fmaf(t.m0, p.x, fmaf(t.m2, p.y, t.t0));
    float py = fmaf(t.m1, p.x, fmaf(t.m3, p.y, t.t1));
    return v2(px, py);
}

// Lines are staged in LDS and leave the workgroup in one coalesced copy (see k_flatten).  `alloc` reserves the n
// slots of one curve piece with ONE LDS atomic; a piece that does not fit the staging area any more goes straight
// to the soup with one global atomic for the piece (bit 31 of the index marks "global").
constexpr uint32_t LINE_IX_GLOBAL = 0x80000000u;
template <uint32_t CAP>
struct FlattenShared {
    uint32_t path_ix[CAP];
    float p0x[CAP], p0y[CAP], p1x[CAP], p1y[CAP];
    uint32_t count;    // slots handed out (may run past the capacity)
    uint32_t lds_end;  // first slot of the first piece that did not fit (dense prefix [0, lds_end) is staged)
    uint32_t base;
};

struct Emitter {
    float bx0, by0, bx1, by1;
    LineSoup *lines;
    uint32_t lines_size;
    Bump *bump;
    uint32_t cap;
    uint32_t *s_path_ix;
    float *s_p0x, *s_p0y, *s_p1x, *s_p1y;
    uint32_t *s_count, *s_lds_end;

    template <uint32_t CAP>
    __device__ __forceinline__ void bind(FlattenShared<CAP> &sh) {
        cap = CAP;
        s_path_ix = sh.path_ix;
        s_p0x = sh.p0x; s_p0y = sh.p0y; s_p1x = sh.p1x; s_p1y = sh.p1y;
        s_count = &sh.count;
        s_lds_end = &sh.lds_end;
    }
    __device__ __forceinline__ uint32_t alloc(uint32_t n) {
        uint32_t slot = atomicAdd(s_count, n);
        if (slot + n <= cap) return slot;
        atomicMin(s_lds_end, slot);
        return atomicAdd(&bump->lines, n) | LINE_IX_GLOBAL;
    }
    // flatten.wgsl:766-773
    __device__ __forceinline__ void write(uint32_t ix, uint32_t path_ix, vec2 p0, vec2 p1) {
        bx0 = minf(bx0, minf(p0.x, p1.x));
        by0 = minf(by0, minf(p0.y, p1.y));
        bx1 = maxf(bx1, maxf(p0.x, p1.x));
        by1 = maxf(by1, maxf(p0.y, p1.y));
        if (ix & LINE_IX_GLOBAL) {
            ix &= ~LINE_IX_GLOBAL;
            if (ix < lines_size) {
                LineSoup l;
                l.path_ix = path_ix; l.pad = 0u;
                l.p0x = p0.x; l.p0y = p0.y; l.p1x = p1.x; l.p1y = p1.y;
                lines[ix] = l;
            }
        } else {
            s_path_ix[ix] = path_ix;
            s_p0x[ix] = p0.x; s_p0y[ix] = p0.y; s_p1x[ix] = p1.x; s_p1y[ix] = p1.y;
        }
    }
    __device__ __forceinline__ void write_xf(uint32_t ix, uint32_t path_ix, vec2 p0, vec2 p1, const Xform &t) {
        write(ix, path_ix, xf_apply(t, p0), xf_apply(t, p1));
    }
    // The cooperative flattener (flatten_euler_coop) writes a line's two ends from different lanes: ONE end of line `ix`
    // (which = 0: p0, 1: p1) and, once per line, its path index.  No box here: the owner lane's box is kept in LDS there.
    __device__ __forceinline__ void write_end(uint32_t ix, uint32_t which, vec2 p) {
        if (ix & LINE_IX_GLOBAL) {
            ix &= ~LINE_IX_GLOBAL;
            if (ix < lines_size) {
                float *d = which ? &lines[ix].p1x : &lines[ix].p0x;
                d[0] = p.x;
                d[1] = p.y;
            }
        } else if (which) {
            s_p1x[ix] = p.x; s_p1y[ix] = p.y;
        } else {
            s_p0x[ix] = p.x; s_p0y[ix] = p.y;
        }
    }
    __device__ __forceinline__ void write_path(uint32_t ix, uint32_t path_ix) {
        if (ix & LINE_IX_GLOBAL) {
            ix &= ~LINE_IX_GLOBAL;
            if (ix < lines_size) {
                lines[ix].path_ix = path_ix;
                lines[ix].pad = 0u;
            }
        } else {
            s_path_ix[ix] = path_ix;
        }
    }
};

// the workgroup's staged lines to their reserved place in the soup: thread i writes the 24-byte record i
template <uint32_t CAP>
__device__ __forceinline__ void copy_staged_lines(const FlattenShared<CAP> &sh, LineSoup *lines, uint32_t lines_size, uint32_t base, uint32_t n_lds,
                                                  uint32_t tid) {
    for (uint32_t i = tid; i < n_lds; i += 256u) {
        const uint32_t o = base + i;
        if (o < lines_size) {
            LineSoup l;
            l.path_ix = sh.path_ix[i]; l.pad = 0u;
            l.p0x = sh.p0x[i]; l.p0y = sh.p0y[i]; l.p1x = sh.p1x[i]; l.p1y = sh.p1y[i];
            lines[o] = l;
        }
    }
}

// ONE atomicAdd(bump.lines) for the staged lines of the workgroup (the reference issues one per line), then a
// coalesced copy: thread i writes the 24-byte record i.  Resets the staging area for the next round.
template <uint32_t CAP>
__device__ __forceinline__ void flush_staged_lines(FlattenShared<CAP> &sh, Bump *bump, LineSoup *lines, uint32_t lines_size, uint32_t tid) {
    __syncthreads();
    const uint32_t n_lds = minu(sh.count, sh.lds_end);
    if (tid == 0u) sh.base = n_lds ? atomicAdd(&bump->lines, n_lds) : 0u;
    __syncthreads();
    copy_staged_lines(sh, lines, lines_size, sh.base, n_lds, tid);
    __syncthreads();
    if (tid == 0u) {
        sh.count = 0u;
        sh.lds_end = 0xffffffffu;
    }
    __syncthreads();
}

// The straight-segment test of flatten_euler (see there), shared with the light kernel.
__device__ __forceinline__ bool cubic_is_straight(vec2 p0, vec2 p1, vec2 p2, vec2 p3, float scale, float offset) {
    vec2 q0 = p1 - p0, q1 = p3 - p2, chd = p3 - p0;
    float c2 = dot(chd, chd), a0 = dot(q0, q0), a1 = dot(q1, q1);
    float h0x = dot(q0, chd), h0y = q0.y * chd.x - q0.x * chd.y;
    float h1x = dot(q1, chd), h1y = q1.x * chd.y - q1.y * chd.x;
    float clen = sqrtf(c2);
    float S = scale * clen;
    float ay0 = fabsf(h0y), ay1 = fabsf(h1y), aoff = fabsf(offset);
    // c2 <= 1e19: beyond it |h|^2 overflows f32 inside the general loop's length(h0) (flatten.wgsl:106-110), which then
    // rejects the range and subdivides where this test would say "one line"; such chords (> 3e9 px) take the loop
    return c2 >= 4e-12f && c2 <= 1e19f && a0 >= 4e-12f && a1 >= 4e-12f && a0 <= c2 && a1 <= c2 && h0x > 0.0f && h1x > 0.0f && ay0 * S <= 0.02f * h0x &&
           ay1 * S <= 0.02f * h1x && aoff * ay0 <= 2.5e-3f * h0x * clen && aoff * ay1 <= 2.5e-3f * h1x * clen;
}

// (tag helpers: no transcendentals)
// unpack2x16float()[0]; vello_encoding/src/math.rs:127-150
__device__ float f16_to_f32(uint32_t bits) {
    const uint32_t MAGIC = 113u << 23;
    const uint32_t SHIFTED_EXP = 0x7c00u << 13;
    uint32_t o = (bits & 0x7fffu) << 13;
    uint32_t e = SHIFTED_EXP & o;
    o += (127u - 15u) << 23;
    if (e == SHIFTED_EXP) {
        o += (128u - 16u) << 23;
    } else if (e == 0u) {
        o += 1u << 23;
        o = __float_as_uint(__uint_as_float(o) - __uint_as_float(MAGIC));
    }
    return __uint_as_float(o | ((bits & 0x8000u) << 16));
}

struct PathTagData {
    uint32_t tag_byte;
    TagMonoid monoid;
};

__device__ __forceinline__ TagMonoid reduce_tag_f(uint32_t tag_word) {
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

// flatten.wgsl:684-701
// (the two loads of compute_tag_monoid apart from its arithmetic: the stroke workgroups request a round's tag words and monoids while the
// round before it is still being flattened)
__device__ __forceinline__ PathTagData tag_monoid_of(uint32_t tag_word, const TagMonoid &pre, uint32_t ix) {
    uint32_t shift = (ix & 3u) * 8u;
    TagMonoid tm = reduce_tag_f(tag_word & ((1u << shift) - 1u));
    PathTagData r;
    r.tag_byte = (tag_word >> shift) & 0xffu;
    r.monoid.trans_ix = pre.trans_ix + tm.trans_ix - 1u;
    r.monoid.pathseg_ix = pre.pathseg_ix + tm.pathseg_ix;
    r.monoid.pathseg_offset = pre.pathseg_offset + tm.pathseg_offset;
    r.monoid.style_ix = pre.style_ix + tm.style_ix - STYLE_SIZE_IN_WORDS;
    r.monoid.path_ix = pre.path_ix + tm.path_ix;
    return r;
}
__device__ PathTagData compute_tag_monoid(const Config &cfg, const uint32_t *scene, const TagMonoid *tag_monoids, uint32_t ix) {
    uint32_t tag_word = scene[cfg.layout.path_tag_base + (ix >> 2)];
    uint32_t shift = (ix & 3u) * 8u;
    TagMonoid tm = reduce_tag_f(tag_word & ((1u << shift) - 1u));
    TagMonoid pre = tag_monoids[ix >> 2];
    PathTagData r;
    r.tag_byte = (tag_word >> shift) & 0xffu;
    r.monoid.trans_ix = pre.trans_ix + tm.trans_ix - 1u;
    r.monoid.pathseg_ix = pre.pathseg_ix + tm.pathseg_ix;
    r.monoid.pathseg_offset = pre.pathseg_offset + tm.pathseg_offset;
    r.monoid.style_ix = pre.style_ix + tm.style_ix - STYLE_SIZE_IN_WORDS;
    r.monoid.path_ix = pre.path_ix + tm.path_ix;
    return r;
}

__device__ __forceinline__ vec2 read_f32_point(const uint32_t *pd, uint32_t ix) {
    return v2(__uint_as_float(pd[ix]), __uint_as_float(pd[ix + 1u]));
}
__device__ __forceinline__ vec2 read_i16_point(const uint32_t *pd, uint32_t ix) {
    uint32_t raw = pd[ix];
    float x = (float)(((int32_t)(raw << 16)) >> 16);
    float y = (float)(((int32_t)raw) >> 16);
    return v2(x, y);
}

// flatten.wgsl:710-764
__device__ CubicPoints read_path_segment(const uint32_t *pd, const PathTagData &tag, bool is_stroke) {
    vec2 p0, p1, p2 = v2(0.0f, 0.0f), p3 = v2(0.0f, 0.0f);
    uint32_t seg_type = tag.tag_byte & PATH_TAG_SEG_TYPE;
    uint32_t off = tag.monoid.pathseg_offset;
    bool is_stroke_cap_marker = is_stroke && (tag.tag_byte & PATH_TAG_SUBPATH_END) != 0u;
    bool is_open = seg_type == PATH_TAG_QUADTO;
    if ((tag.tag_byte & PATH_TAG_F32) != 0u) {
        p0 = read_f32_point(pd, off);
        p1 = read_f32_point(pd, off + 2u);
        if (seg_type >= PATH_TAG_QUADTO) {
            p2 = read_f32_point(pd, off + 4u);
            if (seg_type == PATH_TAG_CUBICTO) p3 = read_f32_point(pd, off + 6u);
        }
    } else {
        p0 = read_i16_point(pd, off);
        p1 = read_i16_point(pd, off + 1u);
        if (seg_type >= PATH_TAG_QUADTO) {
            p2 = read_i16_point(pd, off + 2u);
            if (seg_type == PATH_TAG_CUBICTO) p3 = read_i16_point(pd, off + 3u);
        }
    }
    if (is_stroke_cap_marker && is_open) {
        p0 = p1;
        p1 = p2;
        seg_type = PATH_TAG_LINETO;
    }
    const float third = 1.0f / 3.0f;
    if (seg_type == PATH_TAG_LINETO) {
        p3 = p1;
        p2 = p3 + (p0 - p3) * third;
        p1 = p0 + (p3 - p0) * third;
    } else if (seg_type == PATH_TAG_QUADTO) {
        p3 = p2;
        p2 = p1 + (p2 - p1) * third;
        p1 = p1 + (p0 - p1) * third;
    }
    return CubicPoints{p0, p1, p2, p3};
}

namespace inl {
#define M_UNIT
#define M_SINCOS(x, s, c) sincos_cr(x, s, c)
#define M_SIN(x) sin_cr(x)
#define M_ATAN2(y, x) atan2_cr(y, x)
#define M_ASIN(x) asin_cr(x)
#define M_ACOS(x) acos_cr(x)
#define M_ARC_MATH(x, m) arc_math_inline(x, m)
#include "flatten_walk.inc"
#undef M_UNIT
#undef M_SINCOS
#undef M_SIN
#undef M_ATAN2
#undef M_ASIN
#undef M_ACOS
#undef M_ARC_MATH
}  // namespace inl
namespace outl {
#define M_UNIT __attribute__((noinline))
#define M_SINCOS(x, s, c) fl_sincos(x, s, c)
#define M_SIN(x) fl_sin(x)
#define M_ATAN2(y, x) fl_atan2(y, x)
#define M_ASIN(x) fl_asin(x)
#define M_ACOS(x) fl_acos(x)
#define M_ARC_MATH(x, m) fl_arc_math(x, m)
#include "flatten_walk.inc"
#undef M_UNIT
#undef M_SINCOS
#undef M_SIN
#undef M_ATAN2
#undef M_ASIN
#undef M_ACOS
#undef M_ARC_MATH
}  // namespace outl

// ---- the wave-cooperative form of flatten_euler (round 5) ------------------------------------------------------------------
// flatten_euler above is one lane walking one curve: the subdivision loop and, per accepted range and side, a loop over the
// range's lines with two fp64 sincos and an inverse integral per line.  Lanes of a wave hold DIFFERENT curves, so the wave runs
// the union of all those loops: on mmark-50k the line loop was 45 % of the heavy waves' cycles at 10 % lane use, on the tiger
// (a wave per curve) the launch was as long as its longest curve's serial chain (profiles/r05_flatten_prof_start.txt).
// A line of an accepted range depends only on the range's parameters and its number i (flatten.wgsl:447-470): here the lanes
// walk their subdivisions in lockstep, one turn per pass of the loop, and after every turn ALL 64 lanes flatten the lines of the
// ranges accepted in that turn -- a POINT per lane (point i + 1 of a range is the end of line i and the start of line i + 1),
// the range's parameters read out of the owning lane's registers (ds_bpermute).  Same operations on the same values per point as
// flatten_euler, so the lines are bit-identical; what changes is who computes them and the order they land in the (unordered) soup.
// The owner's box (flatten.wgsl:766-773) is kept in LDS as order-preserving integers (min / max commute); the end point of a
// side's newest range comes back to the owner through LDS (it is the next range's first point).
// Lanes whose inputs are not finite or absurdly large walk flatten_euler instead: NaNs make min / max depend on the order of the
// operands, and the extreme-coordinate tests hold the boxes to the oracle's bit for bit.
constexpr uint32_t EC_PIECES = 32u;  // ranges flattened per pass (a turn in which more lanes accept one takes two passes)
struct EulerPiece {  // what the lines of an accepted range are made of (both sides), written by its owner, read by every lane
    float p0x, p0y, p1x, p1y;        // the range's end points on the cubic (this_p0, this_p1)
    float th0, k0, k1, ch;           // EulerParams
    float noff;                      // offset / chord length of side 0 (side 1: its negative)
    float a[2], b[2], integral[2], int0[2], n[2];
    uint32_t line[2], nu[2];         // first line index, number of lines
    uint32_t flags;                  // robust of side 0 | side 1 << 2 | is_last << 4 | (offset >= 0) << 5 | (-offset >= 0) << 6 | owner lane << 8
    uint32_t first;                  // number of the range's first point among the pass's points
};
struct EulerCoopLds {
    uint32_t bbox[64][4];     // per owner lane: min x, min y, max x, max y of every point written for it (f32_ordered)
    float endpt[64][2][2];    // per owner lane and side: the last point of the newest range (local coordinates)
    EulerPiece piece[EC_PIECES];
    uint32_t lp_stack[32];    // level-parallel walk: the ranges still to be looked at, (level << 16 | range number at that level); the top is next
    uint32_t lp_end_at[36];   // ... which lane's leaf ends at position p (in 32nds of the turn's range)
};
__device__ __forceinline__ uint32_t f32_ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }
__device__ __forceinline__ float shfl_f(float v, uint32_t src) { return __uint_as_float(wave_shfl(__float_as_uint(v), src)); }
#ifdef VELLO_SIMT_EMU
#define FL_WAVE_ANY(c) (__ballot(c) != 0ull)
#else
#define FL_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#endif

// Called by ALL 64 lanes of a wave (valid = this lane has a curve).  Arguments as flatten_euler's.
__device__ void flatten_euler_coop(Emitter &em, EulerCoopLds &cl, bool valid, const CubicPoints &cubic, uint32_t path_ix, const Xform &local_to_device,
                                   float offset, vec2 start_p, vec2 end_p, bool two_sided, vec2 start_n, vec2 end_n, uint32_t lane, bool few_entries) {
    using namespace outl;  // (the walk's pieces with their transcendentals out of line: flatten_walk.inc)
    vec2 p0 = v2(0.0f, 0.0f), p1 = p0, p2 = p0, p3 = p0;
    float scale = 1.0f;
    Xform transform = Xform{1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};
    vec2 t_start[2] = {start_p, start_n}, t_end[2] = {end_p, end_n};
    bool active = valid, tame = true;
    if (valid) {
        if (offset == 0.0f) {
            p0 = xf_apply(local_to_device, cubic.p0);
            p1 = xf_apply(local_to_device, cubic.p1);
            p2 = xf_apply(local_to_device, cubic.p2);
            p3 = xf_apply(local_to_device, cubic.p3);
            t_start[0] = p0;
            t_end[0] = p3;
        } else {
            p0 = cubic.p0; p1 = cubic.p1; p2 = cubic.p2; p3 = cubic.p3;
            transform = local_to_device;
            scale = 0.5f * (length(v2(transform.m0 + transform.m3, transform.m1 - transform.m2)) +
                            length(v2(transform.m0 - transform.m3, transform.m1 + transform.m2)));
        }
        // (NaN fails every comparison: it is "not tame" too)
        const float LIM = 1.0e15f;
        tame = fabsf(p0.x) <= LIM && fabsf(p0.y) <= LIM && fabsf(p1.x) <= LIM && fabsf(p1.y) <= LIM && fabsf(p2.x) <= LIM &&
               fabsf(p2.y) <= LIM && fabsf(p3.x) <= LIM && fabsf(p3.y) <= LIM && fabsf(offset) <= LIM && fabsf(transform.m0) <= LIM &&
               fabsf(transform.m1) <= LIM && fabsf(transform.m2) <= LIM && fabsf(transform.m3) <= LIM && fabsf(transform.t0) <= LIM &&
               fabsf(transform.t1) <= LIM && fabsf(t_start[0].x) <= LIM && fabsf(t_start[0].y) <= LIM && fabsf(t_end[0].x) <= LIM &&
               fabsf(t_end[0].y) <= LIM && fabsf(t_start[1].x) <= LIM && fabsf(t_start[1].y) <= LIM && fabsf(t_end[1].x) <= LIM &&
               fabsf(t_end[1].y) <= LIM;
    }
    // Which waves walk together: a pass over the points of a turn costs ~2.6 line evaluations (who owns the point, 25 words of the
    // range out of LDS, the ends of two lines and the owner's box through LDS) whoever takes part, so it pays where a lane alone would
    // loop long -- a wave with a handful of curves (the tiger: one), or a wave of stroked curves (two sides, a dozen lines a turn:
    // mmark) -- and not for a wave full of fills' curves with two or three lines a range.  Wave-uniform; the lanes with untame
    // inputs walk alone in any case.
    {
        const unsigned long long m_valid = __ballot(valid), m_two = __ballot(valid && two_sided);
#ifdef VK_FL_NO_COOP  // (sweep switch: every lane on its own, always)
        const bool alone = true;
#else
        const bool alone = !few_entries && (uint32_t)__popcll(m_two) * 2u < (uint32_t)__popcll(m_valid);
#endif
        if (valid && (alone || !tame)) {
            flatten_euler(em, cubic, path_ix, local_to_device, offset, start_p, end_p, two_sided, start_n, end_n);
            active = false;
        }
        if (alone) return;
    }
    if (active) {
        if (p0.x == p1.x && p0.y == p1.y && p0.x == p2.x && p0.y == p2.y && p0.x == p3.x && p0.y == p3.y) {
            active = false;
        } else if (cubic_is_straight(p0, p1, p2, p3, scale, offset)) {
            // the straight-segment shortcut, as flatten_euler takes it
            const uint32_t n_sides_u = two_sided ? 2u : 1u;
            const uint32_t line_ix = em.alloc(n_sides_u);
            for (uint32_t side = 0; side < n_sides_u; side++) {
                const float off = side ? -offset : offset;
                vec2 l0 = off >= 0.0f ? t_start[side] : t_end[side];
                vec2 l1 = off >= 0.0f ? t_end[side] : t_start[side];
                em.write_xf(line_ix + side, path_ix, l0, l1, transform);
            }
            flp_mark(FLP_STRAIGHT);
            active = false;
        }
    }
    if (!FL_WAVE_ANY(active)) return;
    const float tol = 0.25f;
    // the owner's box of the points other lanes write for it
    wave_lds_sync();
    cl.bbox[lane][0] = f32_ordered(1e31f); cl.bbox[lane][1] = f32_ordered(1e31f);
    cl.bbox[lane][2] = f32_ordered(-1e31f); cl.bbox[lane][3] = f32_ordered(-1e31f);
    wave_lds_sync();
    const bool walked = active;
    uint32_t t0_u = 0u;
    float dt = 1.0f;
    vec2 last_p = p0;
    vec2 last_q = p1 - p0;
    if (active && dot(last_q, last_q) < DERIV_THRESH_SQUARED) last_q = eval_cubic_and_deriv(p0, p1, p2, p3, DERIV_EPS).deriv;
    float last_t = 0.0f;
    vec2 lp0[2] = {t_start[0], t_start[1]};
    uint32_t off_fwd = (offset >= 0.0f ? 1u : 0u) | ((-offset) >= 0.0f ? 2u : 0u);  // `off >= 0` of side 0 / side 1
    // ---- A wave with ONE curve (the tiger: a wave per curve) walks it LEVEL-PARALLEL.  In lockstep such a wave still pays a turn per
    // range it tests -- the tiger's launch was as long as its longest curve's chain of turns -- but whether a range [t0, t0 + dt] is
    // accepted is a pure function of the range as long as no DERIV_EPS nudge fires (its end points and derivatives are the cubic's at
    // t0 and t0 + dt: what the previous range left in last_p / last_q is that same evaluation), and the ranges the walk accepts are
    // exactly those that pass while every dyadic ancestor fails.  So lane j of the wave takes node j of the dyadic tree under the
    // range on top of a small stack -- five levels, 63 ranges at once -- and one turn decides them all: the accepted leaves up to the
    // first range that needs more than five levels are flattened in ONE pass (each leaf's first point is its predecessor's last,
    // through LDS); that range goes back on the stack for a turn of its own, in front of the blocks to its right.  A tiny derivative
    // anywhere in a turn sends the curve to the lockstep walk from where it stands (the reference's nudges depend on the order).
    const unsigned long long m_active = __ballot(active);
    bool lp_mode = few_entries && __popcll(m_active) == 1;
    const uint32_t lp_owner = lp_mode ? (uint32_t)__ffsll((long long)m_active) - 1u : 0u;
    uint32_t lp_sp = 0u;  // (wave-uniform)
    if (lp_mode) {
        // every lane holds the curve
        p0 = v2(shfl_f(p0.x, lp_owner), shfl_f(p0.y, lp_owner)); p1 = v2(shfl_f(p1.x, lp_owner), shfl_f(p1.y, lp_owner));
        p2 = v2(shfl_f(p2.x, lp_owner), shfl_f(p2.y, lp_owner)); p3 = v2(shfl_f(p3.x, lp_owner), shfl_f(p3.y, lp_owner));
        scale = shfl_f(scale, lp_owner);
        offset = shfl_f(offset, lp_owner);
        two_sided = wave_shfl(two_sided ? 1u : 0u, lp_owner) != 0u;
        off_fwd = wave_shfl(off_fwd, lp_owner);
        path_ix = wave_shfl(path_ix, lp_owner);
        transform.m0 = shfl_f(transform.m0, lp_owner); transform.m1 = shfl_f(transform.m1, lp_owner); transform.m2 = shfl_f(transform.m2, lp_owner);
        transform.m3 = shfl_f(transform.m3, lp_owner); transform.t0 = shfl_f(transform.t0, lp_owner); transform.t1 = shfl_f(transform.t1, lp_owner);
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
            t_end[sd] = v2(shfl_f(t_end[sd].x, lp_owner), shfl_f(t_end[sd].y, lp_owner));
            lp0[sd] = v2(shfl_f(lp0[sd].x, lp_owner), shfl_f(lp0[sd].y, lp_owner));
        }
        last_q = v2(shfl_f(last_q.x, lp_owner), shfl_f(last_q.y, lp_owner));  // (the derivative at t = 0 as the walk starts with it)
        if (lane == 0u) cl.lp_stack[0] = 0u;  // the whole curve: level 0, range 0
        lp_sp = 1u;
        active = false;
        wave_lds_sync();
    }
    for (;;) {
        bool push = false, is_last = false;
        uint32_t pred = 64u;  // the lane whose range ends where this lane's begins (its last point is this range's first); 64: lp0
        uint32_t lp_last = 64u;
        vec2 this_p0 = v2(0.0f, 0.0f), this_p1 = this_p0;
        EulerParams ep{0.0f, 0.0f, 0.0f, 1.0f};
        float noff = 0.0f;
        float s_a[2] = {0.0f, 0.0f}, s_b[2] = {0.0f, 0.0f}, s_integral[2] = {0.0f, 0.0f}, s_int0[2] = {0.0f, 0.0f}, s_n[2] = {1.0f, 1.0f};
        uint32_t s_robust[2] = {0u, 0u}, s_nu[2] = {0u, 0u}, s_line[2] = {0u, 0u};
        // an accepted range [this_p0, this_p1] of chord parameters cp: its Euler segment, per side the integrals and the number of
        // lines, the lines' places (flatten.wgsl:421-446)
        auto accept = [&](const CubicParams &cp) {
            ep = es_params_from_angles(cp.th0, cp.th1);
            const float k0 = ep.k0 - 0.5f * ep.k1;
            const float k1 = ep.k1;
            const float scale_multiplier = sqrtf(0.125f * scale * cp.chord_len / (ep.ch * tol));
            noff = offset / cp.chord_len;
#pragma unroll
            for (int side = 0; side < 2; side++) {
                if (side == 1 && !two_sided) break;
                const float off = side ? -offset : offset;
                const float normalized_offset = off / cp.chord_len;
                const float dist_scaled = normalized_offset * ep.ch;
                float a = 0.0f, b = 0.0f, integral = 0.0f, int0 = 0.0f, n_frac;
                uint32_t robust = ESPC_ROBUST_NORMAL;
                if (fabsf(k1) < K1_THRESH) {
                    const float k = ep.k0;
                    n_frac = sqrtf(fabsf(k * (k * dist_scaled + 1.0f)));
                    robust = ESPC_ROBUST_LOW_K1;
                } else if (fabsf(dist_scaled) < DIST_THRESH) {
                    a = k1;
                    b = k0;
                    int0 = pow_1_5_signed(b);
                    const float int1 = pow_1_5_signed(a + b);
                    integral = int1 - int0;
                    n_frac = (2.0f / 3.0f) * integral / a;
                    robust = ESPC_ROBUST_LOW_DIST;
                } else {
                    a = -2.0f * dist_scaled * k1;
                    b = -1.0f - 2.0f * dist_scaled * k0;
                    int0 = espc_int_approx(b);
                    const float int1 = espc_int_approx(a + b);
                    integral = int1 - int0;
                    const float k_peak = k0 - k1 * b / a;
                    const float integrand_peak = sqrtf(fabsf(k_peak * (k_peak * dist_scaled + 1.0f)));
                    n_frac = integral * integrand_peak / a;
                }
                const float n = clampf(ceilf(n_frac * scale_multiplier), 1.0f, 100.0f);
                s_a[side] = a; s_b[side] = b; s_integral[side] = integral; s_int0[side] = int0; s_n[side] = n;
                s_robust[side] = robust;
                s_nu[side] = f2u(n);
                s_line[side] = em.alloc(s_nu[side]);
                flp_count(FLC_PIECES, 1u);
                flp_count(FLC_EULER_LINES, s_nu[side]);
            }
            flp_mark(FLP_PIECE);
            push = true;
        };
        if (lp_mode) {
            if (lp_sp == 0u) break;  // the curve is done
            // ---- one level-parallel turn: the range on top of the stack and its descendants down five levels ----
            lp_sp -= 1u;
            const uint32_t top = cl.lp_stack[lp_sp];  // (one address for the wave)
            const uint32_t r_level = top >> 16, r_t0u = top & 0xffffu;
            const uint32_t j = lane;  // node j of the tree (1 = the range itself; lane 0 has none)
            const uint32_t d = j ? 31u - (uint32_t)__builtin_clz(j) : 0u, k = j ? j - (1u << d) : 0u;
            const uint32_t lvl = r_level + d;
            const bool node = j != 0u && lvl <= 16u;  // (a range of 2^-16 is accepted whatever it looks like: nothing below it)
            float dt_j = 1.0f;
            for (uint32_t q = 0; q < lvl; q++) dt_j *= 0.5f;  // (as the walk halves it: exact)
            const uint32_t t0u_j = (r_t0u << d) + k;
            const float ts = (float)t0u_j * dt_j;
            const float te = ts + dt_j;
            PointDeriv qa, qb = eval_cubic_and_deriv(p0, p1, p2, p3, te);
            if (t0u_j == 0u) {
                qa.point = p0;      // (the walk starts from p0 itself and from last_q as set up above, not from an evaluation at 0)
                qa.deriv = last_q;
            } else {
                qa = eval_cubic_and_deriv(p0, p1, p2, p3, ts);
            }
            const bool tiny = node && ((t0u_j != 0u && dot(qa.deriv, qa.deriv) < DERIV_THRESH_SQUARED) || dot(qb.deriv, qb.deriv) < DERIV_THRESH_SQUARED);
            if (FL_WAVE_ANY(tiny)) {
                // a nudge somewhere under this range: the owner walks on from the range's start, alone, in lockstep's code
                lp_mode = false;
                active = lane == lp_owner;
                t0_u = r_t0u;
                dt = 1.0f;
                for (uint32_t q = 0; q < r_level; q++) dt *= 0.5f;
                last_t = (float)t0_u * dt;
                if (t0_u == 0u) {
                    last_p = p0;  // (last_q: as set up above)
                } else {
                    const PointDeriv st = eval_cubic_and_deriv(p0, p1, p2, p3, last_t);
                    last_p = st.point;
                    last_q = st.deriv;
                }
                continue;
            }
            const CubicParams cp = cubic_from_points_derivs(qa.point, qb.point, qa.deriv, qb.deriv, te - ts);
            flp_mark(FLP_SUBDIV);
            flp_count(FLC_ITERS, node ? 1u : 0u);
            const bool pass = node && (cp.err * scale <= tol || dt_j <= SUBDIV_LIMIT);
            const unsigned long long m_pass = __ballot(pass);
            bool anc_pass = false;
            for (uint32_t a = j >> 1; a >= 1u; a >>= 1) anc_pass = anc_pass || ((m_pass >> a) & 1ull) != 0ull;
            const bool leaf = pass && !anc_pass;
            const bool unres = node && d == 5u && !pass && !anc_pass;
            const unsigned long long m_unres = __ballot(unres);  // (lanes 32 .. 63 are the level-5 ranges in t order)
            const uint32_t u_lane = m_unres ? (uint32_t)__ffsll((long long)m_unres) - 1u : 64u;
            const uint32_t limit = m_unres ? u_lane - 32u : 32u;  // in 32nds of the range: what lies before the first unresolved range
            const uint32_t span32 = 32u >> d, start_pos = k * span32, end_pos = start_pos + span32;
            const bool take = leaf && end_pos <= limit;
            // who ends where (a leaf's first point is the last point of the leaf that ends where it begins)
            wave_lds_sync();
            if (take) cl.lp_end_at[end_pos] = lane;
            wave_lds_sync();
            if (take && start_pos != 0u) pred = cl.lp_end_at[start_pos];
            if (limit != 0u) lp_last = cl.lp_end_at[limit];  // (all of [0, limit) is leaves: one of them ends at limit)
            if (take) {
                this_p0 = qa.point;
                this_p1 = qb.point;
                is_last = te == 1.0f;
                accept(cp);
            }
            // the stack: the unresolved range on top, the blocks to its right below it, the leftmost of them uppermost
            if (m_unres != 0ull) {
                const uint32_t ku = u_lane - 32u;
                if (lane == 0u) {
                    uint32_t sp = lp_sp;
                    for (uint32_t dd = 1u; dd <= 5u; dd++) {
                        const uint32_t a = ku >> (5u - dd);  // the unresolved range's ancestor at depth dd (itself at 5)
                        if ((a & 1u) == 0u) cl.lp_stack[sp++] = ((r_level + dd) << 16) | ((r_t0u << dd) + a + 1u);
                    }
                    cl.lp_stack[sp++] = ((r_level + 5u) << 16) | ((r_t0u << 5) + ku);
                }
                uint32_t n_push = 1u;
                for (uint32_t dd = 1u; dd <= 5u; dd++) n_push += ((ku >> (5u - dd)) & 1u) == 0u ? 1u : 0u;
                lp_sp += n_push;
                wave_lds_sync();
            }
        } else {
            if (!FL_WAVE_ANY(active)) break;
            // ---- one turn of every active lane's walk (flatten.wgsl:395-446) ----
            if (active) {
                const float t0 = (float)t0_u * dt;
                if (t0 == 1.0f) {
                    active = false;
                } else {
                    float t1 = t0 + dt;
                    PointDeriv this_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1);
                    if (dot(this_pq1.deriv, this_pq1.deriv) < DERIV_THRESH_SQUARED) {
                        PointDeriv new_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1 - DERIV_EPS);
                        this_pq1.deriv = new_pq1.deriv;
                        if (t1 < 1.0f) {
                            this_pq1.point = new_pq1.point;
                            t1 = t1 - DERIV_EPS;
                        }
                    }
                    const float actual_dt = t1 - last_t;
                    const CubicParams cp = cubic_from_points_derivs(last_p, this_pq1.point, last_q, this_pq1.deriv, actual_dt);
                    flp_mark(FLP_SUBDIV);
                    flp_count(FLC_ITERS, 1u);
                    if (cp.err * scale <= tol || dt <= SUBDIV_LIMIT) {
                        this_p0 = last_p;
                        this_p1 = this_pq1.point;
                        is_last = t1 == 1.0f;
                        accept(cp);
                        last_p = this_pq1.point;
                        last_q = this_pq1.deriv;
                        last_t = t1;
                        t0_u += 1u;
                        const uint32_t shift = (uint32_t)(__ffs((int)t0_u) - 1);
                        t0_u >>= shift;
                        dt *= (float)(1u << shift);
                    } else {
                        t0_u = t0_u * 2u;
                        dt *= 0.5f;
                    }
                }
            }
        }
        unsigned long long pending = __ballot(push);
        if (pending == 0ull) continue;
        // ---- the lines of the ranges accepted in this turn, a point per lane (flatten.wgsl:447-470): the owners leave their ranges in
        // LDS (EC_PIECES at a time, by rank among the accepting lanes), every lane takes points ----
        const uint32_t box_lp = lp_mode ? lp_owner : 64u;  // (level-parallel: every range is the one curve's)
        while (pending != 0ull) {
            const uint32_t rank = mask_rank_below(pending, lane);
            const bool mine = push && ((pending >> lane) & 1ull) != 0ull && rank < EC_PIECES;
            const uint32_t cnt = mine ? s_nu[0] + s_nu[1] : 0u;
            const uint32_t incl = wave_incl_scan_u32(cnt, (int)lane);
            const uint32_t total = wave_read(incl, 63u);
            const uint32_t n_pieces = minu((uint32_t)__popcll(pending), EC_PIECES);
            wave_lds_sync();
            if (mine) {
                EulerPiece &pc = cl.piece[rank];
                pc.p0x = this_p0.x; pc.p0y = this_p0.y; pc.p1x = this_p1.x; pc.p1y = this_p1.y;
                pc.th0 = ep.th0; pc.k0 = ep.k0; pc.k1 = ep.k1; pc.ch = ep.ch;
                pc.noff = noff;
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    pc.a[sd] = s_a[sd]; pc.b[sd] = s_b[sd]; pc.integral[sd] = s_integral[sd]; pc.int0[sd] = s_int0[sd]; pc.n[sd] = s_n[sd];
                    pc.line[sd] = s_line[sd]; pc.nu[sd] = s_nu[sd];
                }
                pc.flags = s_robust[0] | (s_robust[1] << 2) | (is_last ? 16u : 0u) | (off_fwd << 5) | (lane << 8);
                pc.first = incl - cnt;
            }
            wave_lds_sync();
            for (uint32_t base = 0u; base < total; base += 64u) {
                const bool on = base + lane < total;
                const uint32_t q = on ? base + lane : total - 1u;
                // the range: the last one whose first point is <= q (a range has at least one line)
                uint32_t pi = 0u;
#pragma unroll
                for (uint32_t step = EC_PIECES / 2u; step >= 1u; step >>= 1) {
                    const uint32_t probe = pi + step;
                    if (probe < n_pieces && cl.piece[probe].first <= q) pi = probe;
                }
                const EulerPiece &pc = cl.piece[pi];
                const uint32_t local = q - pc.first;
                const uint32_t o_nA = pc.nu[0];
                const uint32_t side = local >= o_nA ? 1u : 0u;
                const uint32_t i = local - (side ? o_nA : 0u);
                const uint32_t e_flags = pc.flags;
                const uint32_t owner = e_flags >> 8;
                const vec2 e_p0 = v2(pc.p0x, pc.p0y), e_p1 = v2(pc.p1x, pc.p1y);
                EulerParams e_ep;
                e_ep.th0 = pc.th0; e_ep.k0 = pc.k0; e_ep.k1 = pc.k1; e_ep.ch = pc.ch;
                const float a = pc.a[side], b = pc.b[side], integral = pc.integral[side], int0 = pc.int0[side], n = pc.n[side];
                const uint32_t n_u = pc.nu[side], line_ix = pc.line[side];
                const uint32_t robust = (e_flags >> (side * 2u)) & 3u;
                const bool e_last = (e_flags & 16u) != 0u, fwd = ((e_flags >> (5u + side)) & 1u) != 0u;
                const float normalized_offset = side ? -pc.noff : pc.noff;
                // (what does not change from turn to turn stays in the owner's registers: the curve's end points, its transform and
                // path; every lane takes part in a ds_bpermute, so they are fetched outside the branches)
                const vec2 te0 = v2(shfl_f(t_end[0].x, owner), shfl_f(t_end[0].y, owner)), te1 = v2(shfl_f(t_end[1].x, owner), shfl_f(t_end[1].y, owner));
                vec2 lp1;
                if (i + 1u == n_u && e_last) {
                    lp1 = side ? te1 : te0;
                } else {
                    const float t = (float)(i + 1u) / n;
                    float sv = t;
                    if (robust != ESPC_ROBUST_LOW_K1) {
                        const float u = integral * t + int0;
                        float inv;
                        if (robust == ESPC_ROBUST_LOW_DIST) inv = pow_cr(fabsf(u), 2.0f / 3.0f) * signf(u);
                        else inv = espc_int_inv_approx(u);
                        sv = (inv - b) / a;
                    }
                    lp1 = es_seg_eval_with_offset(e_p0, e_p1, e_ep, sv, normalized_offset);
                }
                Xform e_t;
                e_t.m0 = shfl_f(transform.m0, owner); e_t.m1 = shfl_f(transform.m1, owner); e_t.m2 = shfl_f(transform.m2, owner);
                e_t.m3 = shfl_f(transform.m3, owner); e_t.t0 = shfl_f(transform.t0, owner); e_t.t1 = shfl_f(transform.t1, owner);
                const uint32_t e_path = wave_shfl(path_ix, owner);
                if (on) {
                    // point i + 1: the far end of line i, the near end of line i + 1 (flatten.wgsl:463-468: the ends swap for a
                    // negative offset); the range's LAST point goes to LDS: the next range's first (and, below, its own line 0 waits
                    // for its predecessor's)
                    const vec2 P = xf_apply(e_t, lp1);
                    em.write_end(line_ix + i, fwd ? 1u : 0u, P);
                    em.write_path(line_ix + i, e_path);
                    if (i + 1u < n_u) {
                        em.write_end(line_ix + i + 1u, fwd ? 0u : 1u, P);
                    } else {
                        cl.endpt[owner][side][0] = lp1.x;
                        cl.endpt[owner][side][1] = lp1.y;
                    }
                    const uint32_t row = box_lp < 64u ? box_lp : owner;
                    atomicMin(&cl.bbox[row][0], f32_ordered(P.x));
                    atomicMin(&cl.bbox[row][1], f32_ordered(P.y));
                    atomicMax(&cl.bbox[row][2], f32_ordered(P.x));
                    atomicMax(&cl.bbox[row][3], f32_ordered(P.y));
                }
                flp_mark(FLP_EMIT);
            }
            // the first EC_PIECES accepting lanes are served
            {
                unsigned long long m = pending;
                for (uint32_t k = 0; k < n_pieces; k++) m &= m - 1ull;
                pending = m;
            }
        }
        wave_lds_sync();
        // every range's first point -- the near end of its line 0: the last point of the range before it (lockstep: this lane's own
        // previous range, lp0; level-parallel: the leaf that ends where this one begins) -- by the lane that accepted the range
        if (push) {
            const uint32_t row = box_lp < 64u ? box_lp : lane;
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                if (sd == 1 && !two_sided) break;
                const vec2 st = pred < 64u ? v2(cl.endpt[pred][sd][0], cl.endpt[pred][sd][1]) : lp0[sd];
                const vec2 S = xf_apply(transform, st);
                const bool fwd = ((off_fwd >> sd) & 1u) != 0u;
                em.write_end(s_line[sd], fwd ? 0u : 1u, S);
                atomicMin(&cl.bbox[row][0], f32_ordered(S.x));
                atomicMin(&cl.bbox[row][1], f32_ordered(S.y));
                atomicMax(&cl.bbox[row][2], f32_ordered(S.x));
                atomicMax(&cl.bbox[row][3], f32_ordered(S.y));
            }
        }
        // ... and where the next range starts
        if (lp_mode) {
            if (lp_last < 64u) {
                lp0[0] = v2(cl.endpt[lp_last][0][0], cl.endpt[lp_last][0][1]);
                if (two_sided) lp0[1] = v2(cl.endpt[lp_last][1][0], cl.endpt[lp_last][1][1]);
            }
        } else if (push) {
            lp0[0] = v2(cl.endpt[lane][0][0], cl.endpt[lane][0][1]);
            if (two_sided) lp0[1] = v2(cl.endpt[lane][1][0], cl.endpt[lane][1][1]);
        }
        wave_lds_sync();
    }
    if (walked) {
        em.bx0 = minf(em.bx0, f32_from_ordered(cl.bbox[lane][0]));
        em.by0 = minf(em.by0, f32_from_ordered(cl.bbox[lane][1]));
        em.bx1 = maxf(em.bx1, f32_from_ordered(cl.bbox[lane][2]));
        em.by1 = maxf(em.by1, f32_from_ordered(cl.bbox[lane][3]));
    }
}

// flatten_tag for a whole wave (has_tag = this lane has a list entry): the same three steps -- decode, the two offset curves (or
// the fill's curve), join or cap -- with the middle one taken by all 64 lanes together (flatten_euler_coop).
__device__ uint32_t flatten_tag_coop(Emitter &em, EulerCoopLds &cl, bool has_tag, const Config &cfg, const uint32_t *scene, const TagMonoid *tag_monoids,
                                     uint32_t ix, uint32_t lane, bool few_entries) {
    using namespace outl;
    em.bx0 = 1e31f; em.by0 = 1e31f; em.bx1 = -1e31f; em.by1 = -1e31f;
    uint32_t path_ix = 0xffffffffu, style_flags = 0u;
    bool euler = false, two_sided = false, start_cap = false, do_join = false, end_cap = false;
    CubicPoints pts{v2(0.0f, 0.0f), v2(0.0f, 0.0f), v2(0.0f, 0.0f), v2(0.0f, 0.0f)};
    Xform transform{1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};
    float offset = 0.0f;
    vec2 e_start_p = v2(0.0f, 0.0f), e_end_p = e_start_p, e_start_n = e_start_p, e_end_n = e_start_p;
    vec2 tan_prev = e_start_p, tan_next = e_start_p, n_prev = e_start_p, n_next = e_start_p, offset_tangent = e_start_p;
    if (has_tag) {
        PathTagData tag = compute_tag_monoid(cfg, scene, tag_monoids, ix);
        const uint32_t seg_type = tag.tag_byte & PATH_TAG_SEG_TYPE;
        path_ix = tag.monoid.path_ix;
        // (a PATH marker's draw flags / transform index, flatten.wgsl:813-817: stored by k_pathtag_scan, scan.hip)
        if (seg_type != 0u) {
            const uint32_t style_ix = tag.monoid.style_ix;
            style_flags = scene[(uint32_t)(cfg.layout.style_base + style_ix)];
            const uint32_t *pd = scene + cfg.layout.path_data_base;
            const bool is_stroke = (style_flags & STYLE_FLAGS_STYLE) != 0u;
            transform = read_transform(scene, cfg.layout.transform_base, tag.monoid.trans_ix);
            pts = read_path_segment(pd, tag, is_stroke);
            flp_count(FLC_ENTRIES, 1u);
            if (is_stroke) {
                const float linewidth = __uint_as_float(scene[cfg.layout.style_base + style_ix + 1u]);
                offset = 0.5f * linewidth;
                const bool is_open = seg_type != PATH_TAG_LINETO;
                const bool is_stroke_cap_marker = (tag.tag_byte & PATH_TAG_SUBPATH_END) != 0u;
                if (is_stroke_cap_marker) {
                    if (is_open) {
                        const vec2 tangent = pts.p3 - pts.p0;
                        offset_tangent = normalize(tangent) * offset;
                        start_cap = true;
                    }
                } else {
                    // read_neighboring_segment(ix + 1), flatten.wgsl:810-822
                    PathTagData ntag = compute_tag_monoid(cfg, scene, tag_monoids, ix + 1u);
                    CubicPoints npts = read_path_segment(pd, ntag, true);
                    const bool n_is_closed = (ntag.tag_byte & PATH_TAG_SEG_TYPE) == PATH_TAG_LINETO;
                    const bool n_is_marker = (ntag.tag_byte & PATH_TAG_SUBPATH_END) != 0u;
                    do_join = !n_is_marker || n_is_closed;
                    end_cap = !do_join;
                    vec2 n_tangent = npts.p3 - npts.p0;
                    if (!n_is_marker) n_tangent = cubic_start_tangent(npts.p0, npts.p1, npts.p2, npts.p3);
                    vec2 tan_start = cubic_start_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                    if (dot(tan_start, tan_start) < TANGENT_THRESH * TANGENT_THRESH) tan_start = v2(TANGENT_THRESH, 0.0f);
                    tan_prev = cubic_end_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                    if (dot(tan_prev, tan_prev) < TANGENT_THRESH * TANGENT_THRESH) tan_prev = v2(TANGENT_THRESH, 0.0f);
                    tan_next = n_tangent;
                    if (dot(tan_next, tan_next) < TANGENT_THRESH * TANGENT_THRESH) tan_next = v2(TANGENT_THRESH, 0.0f);
                    const vec2 n_start = normalize(v2(-tan_start.y, tan_start.x)) * offset;
                    offset_tangent = normalize(tan_prev) * offset;
                    n_prev = v2(-offset_tangent.y, offset_tangent.x);
                    const vec2 tnn = normalize(tan_next) * offset;
                    n_next = v2(-tnn.y, tnn.x);
                    euler = two_sided = true;
                    e_start_p = pts.p0 + n_start; e_end_p = pts.p3 + n_prev; e_start_n = pts.p0 - n_start; e_end_n = pts.p3 - n_prev;
                }
            } else {
                euler = true;
                e_start_p = pts.p0; e_end_p = pts.p3; e_start_n = pts.p0; e_end_n = pts.p3;
            }
        }
    }
    flp_mark(FLP_TAG);
    flatten_euler_coop(em, cl, euler, pts, path_ix, transform, offset, e_start_p, e_end_p, two_sided, e_start_n, e_end_n, lane, few_entries);
    flp_mark(FLP_SUBDIV);  // (the walk's exit, and the lanes that had nothing to walk)
    if (start_cap) {
        const vec2 n = v2(-offset_tangent.y, offset_tangent.x);
        draw_cap(em, path_ix, (style_flags & STYLE_FLAGS_START_CAP_MASK) >> 2, pts.p0, pts.p0 - n, pts.p0 + n, -offset_tangent, transform);
    } else if (do_join) {
        draw_join(em, path_ix, style_flags, pts.p3, tan_prev, tan_next, n_prev, n_next, transform);
    } else if (end_cap) {
        draw_cap(em, path_ix, style_flags & STYLE_FLAGS_END_CAP_MASK, pts.p3, pts.p3 + n_prev, pts.p3 - n_prev, offset_tangent, transform);
    }
    return path_ix;
}

// Per-path bbox update.  The reference issues 4 global atomics per segment tag (flatten.wgsl:916-921); device-
// scope atomics are a scarce resource on MI355X (~2e10/s chip-wide, measured), so the 64 consecutive tags a
// wave holds are first combined with a segmented shuffle scan keyed by path index (keys are non-decreasing in
// tag order), and only the last lane of each run touches memory: ~2-3 atomic groups per wave instead of 64.
// min/max commute with the monotone floor/ceil, so the resulting integer bbox is identical.
__device__ __forceinline__ void wave_bbox_update(PathBbox *path_bboxes, uint32_t n_paths, uint32_t key, float x0, float y0, float x1,
                                                 float y1, int lane) {
    // (a wave without a single extent -- the light kernel's waves of stroked tags, which it only queues -- has nothing to merge and
    // nothing to send: thirty shuffles saved)
    if (!FL_WAVE_ANY(key < n_paths && (x1 > x0 || y1 > y0))) return;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t ok_key = __shfl_up(key, d);
        float ox0 = __shfl_up(x0, d), oy0 = __shfl_up(y0, d), ox1 = __shfl_up(x1, d), oy1 = __shfl_up(y1, d);
        if (lane >= d && ok_key == key) {
            x0 = minf(x0, ox0); y0 = minf(y0, oy0); x1 = maxf(x1, ox1); y1 = maxf(y1, oy1);
        }
    }
    uint32_t next_key = __shfl_down(key, 1);
    bool tail = lane == 63 || next_key != key;
    if (tail && key < n_paths && (x1 > x0 || y1 > y0)) {
        PathBbox *out = &path_bboxes[key];
        atomicMin(&out->x0, f2i(floorf(x0)));
        atomicMin(&out->y0, f2i(floorf(y0)));
        atomicMax(&out->x1, f2i(ceilf(x1)));
        atomicMax(&out->y1, f2i(ceilf(y1)));
    }
}

}  // namespace

// ---- light kernel -------------------------------------------------------------------------------------------
// Most tags of a map-like scene are fill segments that pass the straight-segment test: one transform, one test, one
// line.  Inlined next to the Euler-spiral / stroker code they inherit its 256 VGPRs (2 waves per SIMD), and the
// kernel is bound by the chain of dependent loads tag -> monoid -> style / transform / points with nothing to hide
// it behind.  This kernel handles exactly those tags (and the PATH markers) with a small register budget; every
// other segment tag is queued for the heavy code, curves and strokes on separate lists so that its waves are
// homogeneous (a wave mixing 8 cubics with 56 stroked lines runs the subdivision loop at 1/8 lane use).
// Returns 0 = done, HEAVY_CURVE or HEAVY_STROKE.
constexpr uint32_t HEAVY_CURVE = 1u, HEAVY_STROKE = 2u, HEAVY_STROKE_LINE = 3u;
__device__ __forceinline__ uint32_t flatten_tag_light(Emitter &em, const Config &cfg, const uint32_t *scene, const TagMonoid *tag_monoids,
                                                  PathBbox *path_bboxes, uint32_t ix, uint32_t &path_ix_out) {
    PathTagData tag = compute_tag_monoid(cfg, scene, tag_monoids, ix);
    uint32_t seg_type = tag.tag_byte & PATH_TAG_SEG_TYPE;
    uint32_t path_ix = tag.monoid.path_ix;
    path_ix_out = path_ix;
    em.bx0 = 1e31f; em.by0 = 1e31f; em.bx1 = -1e31f; em.by1 = -1e31f;
    // (a PATH marker's draw flags / transform index, flatten.wgsl:813-817: stored by k_pathtag_scan, scan.hip)
    if (seg_type == 0u) return 0u;
    uint32_t style_ix = tag.monoid.style_ix;
    uint32_t trans_ix = tag.monoid.trans_ix;
    uint32_t style_flags = scene[(uint32_t)(cfg.layout.style_base + style_ix)];
    // joins, caps, offset curves; stroked LINES (most of a map) are listed apart: they need no Euler spirals
    if ((style_flags & STYLE_FLAGS_STYLE) != 0u) return seg_type == PATH_TAG_LINETO ? HEAVY_STROKE_LINE : HEAVY_STROKE;
    const uint32_t *pd = scene + cfg.layout.path_data_base;
    Xform transform = read_transform(scene, cfg.layout.transform_base, trans_ix);
    CubicPoints pts = read_path_segment(pd, tag, false);
    // flatten_euler with offset == 0 (flatten.wgsl:340-352): points to device space, degenerate cubics emit nothing
    vec2 p0 = xf_apply(transform, pts.p0), p1 = xf_apply(transform, pts.p1), p2 = xf_apply(transform, pts.p2),
         p3 = xf_apply(transform, pts.p3);
    if (p0.x == p1.x && p0.y == p1.y && p0.x == p2.x && p0.y == p2.y && p0.x == p3.x && p0.y == p3.y) return 0u;
    if (!cubic_is_straight(p0, p1, p2, p3, 1.0f, 0.0f)) return HEAVY_CURVE;
    const Xform identity{1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f};
    uint32_t line_ix = em.alloc(1u);
    em.write_xf(line_ix, path_ix, p0, p3, identity);  // the reference applies the (identity) transform here too
    return 0u;
}

// (one workgroup of flatten's light pass: FLATTEN_BLOCK_TAGS tags from tag `block` x FLATTEN_BLOCK_TAGS on)
__device__ __forceinline__ void flatten_light_workgroup(const Config &cfg, uint32_t block, uint32_t n_tags, const uint32_t *scene, const TagMonoid *tag_monoids,
                                                        PathBbox *path_bboxes, Control *control, LineSoup *lines, uint32_t *heavy_list) {
    __shared__ FlattenShared<FLATTEN_BLOCK_TAGS> sh;  // at most one line per tag: never overflows
    __shared__ uint32_t sh_heavy[FLATTEN_BLOCK_TAGS];  // curves from the front, strokes from the back
    __shared__ uint32_t sh_lines[FLATTEN_BLOCK_TAGS];  // stroked lines
    __shared__ uint32_t sh_n_heavy[3], sh_heavy_base[3];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    // Lane t of a wave takes tag t of a 64-tag run (4 runs per thread, 256 tags apart): consecutive tags of
    // a path are of one kind, so waves stay convergent and segment reads coalesce.
    const uint32_t tag0 = block * FLATTEN_BLOCK_TAGS + tid;
    if ((control->bump.failed & FAILED_SCENE) != 0u) return;  // the tag stream overruns the scene: nothing may be indexed with it
    if (tid == 0u) {
        sh.count = 0u;
        sh.lds_end = 0xffffffffu;
        sh_n_heavy[0] = sh_n_heavy[1] = sh_n_heavy[2] = 0u;
    }
    __syncthreads();
    Bump *bump = &control->bump;
#pragma unroll 1
    for (uint32_t j = 0; j < FLATTEN_TAGS_PER_THREAD; j++) {
        Emitter em;
        em.lines = lines;
        em.lines_size = cfg.lines_size;
        em.bump = bump;
        em.bind(sh);
        uint32_t ix = tag0 + j * 256u;
        uint32_t key = 0xffffffffu;
        float x0 = 1e31f, y0 = 1e31f, x1 = -1e31f, y1 = -1e31f;
        uint32_t heavy = 0u;
        if (ix < n_tags) {
            heavy = flatten_tag_light(em, cfg, scene, tag_monoids, path_bboxes, ix, key);
            // a tag contributes only if it produced an extent (flatten.wgsl:915)
            if (em.bx1 > em.bx0 || em.by1 > em.by0) {
                x0 = em.bx0; y0 = em.by0; x1 = em.bx1; y1 = em.by1;
            }
        }
        // wave-aggregated appends to the workgroup's two heavy lists (tag order is kept inside a wave)
#pragma unroll
        for (uint32_t kind = 0; kind < 3u; kind++) {
            const unsigned long long hm = __ballot(heavy == kind + 1u);
            if (hm != 0ull) {
                const int leader = __ffsll((long long)hm) - 1;
                uint32_t wbase = 0u;
                if ((int)lane == leader) wbase = atomicAdd(&sh_n_heavy[kind], (uint32_t)__popcll(hm));
                wbase = (uint32_t)__shfl((int)wbase, leader);
                if (heavy == kind + 1u) {
                    uint32_t pos = wbase + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                    if (kind == 2u) sh_lines[pos] = ix;
                    else sh_heavy[kind == 0u ? pos : FLATTEN_BLOCK_TAGS - 1u - pos] = ix;
                }
            }
        }
        wave_bbox_update(path_bboxes, cfg.layout.n_paths, key, x0, y0, x1, y1, (int)lane);
    }
    // The workgroup's four reservations -- its lines in the soup, its entries in the three lists -- as ONE instruction of four
    // lanes: a same-address atomic is a queue of every workgroup of the launch (12 ns each, 933 of them on the road map), and
    // the soup's after the copy-out, then the lists', were two such queues in a row on every workgroup's way out.
    __syncthreads();
    const uint32_t n_lds = minu(sh.count, sh.lds_end);
    if (tid < 4u) {
        uint32_t *const counter = tid == 0u ? &bump->lines : &control->heavy_count[tid - 1u];
        const uint32_t n = tid == 0u ? n_lds : sh_n_heavy[tid - 1u];
        const uint32_t got = n ? atomicAdd(counter, n) : 0u;
        if (tid == 0u) sh.base = got;
        else sh_heavy_base[tid - 1u] = got;
    }
    __syncthreads();
    copy_staged_lines(sh, lines, cfg.lines_size, sh.base, n_lds, tid);
    // curves fill heavy_list[0, n_tags), strokes [n_tags, 2 n_tags), stroked lines [2 n_tags, 3 n_tags) (the stroke workgroups
    // append the lines they hand on to [3 n_tags, 4 n_tags))
    for (uint32_t i = tid; i < sh_n_heavy[0]; i += 256u) heavy_list[sh_heavy_base[0] + i] = sh_heavy[i];
    for (uint32_t i = tid; i < sh_n_heavy[1]; i += 256u) heavy_list[n_tags + sh_heavy_base[1] + i] = sh_heavy[FLATTEN_BLOCK_TAGS - 1u - i];
    for (uint32_t i = tid; i < sh_n_heavy[2]; i += 256u) heavy_list[2u * n_tags + sh_heavy_base[2] + i] = sh_lines[i];
}

// The first n_draw_blocks workgroups are the draw stage's (draw_scan.h; first, so that their look-back chain is under way while
// the flatten workgroups fill the chip): they need the scene and what
// k_pathtag_scan stored at the PATH markers, nothing of flatten's, and the stage as a launch of its own is 6-10 us of launch
// boundary and look-back latency on the frame's critical path.
__global__ void __launch_bounds__(256, 4) k_flatten_light(Config cfg, uint32_t n_tags, const uint32_t *__restrict__ scene,
                                                          const TagMonoid *__restrict__ tag_monoids, PathBbox *path_bboxes,
                                                          Control *control, LineSoup *lines, uint32_t *heavy_list, uint32_t n_draw_blocks,
                                                          unsigned long long *draw_state, DrawMonoid *__restrict__ draw_monoids,
                                                          uint32_t *__restrict__ info, Clip *__restrict__ clip_inp) {
    if (blockIdx.x < n_draw_blocks) {
        draw_scan_workgroup(cfg, scene, control, draw_state, path_bboxes, draw_monoids, info, clip_inp);
        return;
    }
    flatten_light_workgroup(cfg, blockIdx.x - n_draw_blocks, n_tags, scene, tag_monoids, path_bboxes, control, lines, heavy_list);
}

// ---- stroked lines: flatten_tag's stroke branch without the Euler-spiral flattener ------------------------------
// Same operations in the same order as flatten_tag -> flatten_euler's straight-segment shortcut -> draw_join / draw_cap.
// Returns false (having emitted nothing) for a line flatten_euler would not take that shortcut for (degenerate, or offset
// lines that fail the straight-segment test): the caller queues it for the heavy code.
// An arc of a round join or cap whose lines cannot be told without the exact transcendentals (atan2, acos, sincos in fp64):
// set aside by the lane of the stroke workgroup that met it and flattened by a lane of its own of the heavy code, where such
// arcs are dense (a few per cent of a road map's stroked lines: inline, nearly every wave walked the fp64 routines for its
// one or two of them).  16 words; the workgroup collects its arcs in LDS and appends them to the frame's list behind ONE
// atomic.
constexpr uint32_t STROKE_ARCS = 256u;  // at most one per stroked line of a round
constexpr uint32_t ARC_IS_CAP = 1u;
struct __attribute__((aligned(16))) ArcItem {
    uint32_t path_ix, trans_ix, flags, pad;
    float bx, by, ex, ey;   // begin, end
    float cx, cy, cr, d;    // center; the angle is |atan2(cr, d)| (pi for a cap)
    float box[4];           // extent of the OTHER lines of the thread that met the arc: the thread's box is tested and merged as one
};
static_assert(sizeof(ArcItem) == 64, "ArcItem");
struct ArcQueue {
    ArcItem item[STROKE_ARCS];
    uint32_t count, base;
};

// Round join or cap between `begin` and `end` about `center` (flatten_arc's arguments; the angle is |atan2(cr, d)|, pi for
// a cap).  flatten_arc makes n = max(1, ceil(angle / theta)) lines with theta = 2 acos(1 - tol / radius): for the joins of
// a road map n is 1 nearly always, and then neither the angle nor theta nor sincos(theta) is needed, only the fact.
// angle < theta  <=>  cos(angle) > cos(theta) on [0, pi], with cos(angle) = d / |(cr, d)| and cos(theta) = 2 x^2 - 1:
// one sqrt and one division.  The margin (1e-4 in the cosine: >= 3e-5 relative in the quotient angle / theta, against
// < 1e-6 of rounding in everything the reference computes in f32) makes the shortcut safe, not tight: what it does not
// decide goes to the queue and is flattened exactly.  Returns true when the arc was emitted here.
__device__ __forceinline__ bool stroke_arc_one_line(Emitter &em, uint32_t path_ix, vec2 begin, vec2 end, vec2 center, float cr, float d,
                                                    const Xform &transform) {
    const float tol = 0.25f;
    const vec2 p0 = xf_apply(transform, begin);
    const float radius = maxf(tol, length(p0 - xf_apply(transform, center)));
    const float x = 1.0f - tol / radius;
    const float cos_theta = 2.0f * x * x - 1.0f;
    const float cos_angle = d / sqrtf(cr * cr + d * d);
    if (!(cos_angle > cos_theta + 1.0e-4f)) return false;  // (NaNs land here too)
    const uint32_t ix = em.alloc(1u);
    em.write(ix, path_ix, p0, xf_apply(transform, end));
    return true;
}

// One stroked LINETO (flatten_tag's stroke branch with flatten_euler reduced to its straight-segment shortcut).  Returns
// false when the segment is not straight (the heavy kernel takes it).  Arcs that need the exact path are pushed to `q`.
// tag_word / pre: the tag word ix >> 2 and its monoid, loaded by the caller (a round ahead).
__device__ bool flatten_stroked_line(Emitter &em, ArcQueue &q, const Config &cfg, const uint32_t *scene, const TagMonoid *tag_monoids,
                                     uint32_t ix, uint32_t tag_word, const TagMonoid &pre, uint32_t &path_ix_out) {
    using namespace inl;  // (tangents, joins and caps without arcs: no transcendentals either way)
    PathTagData tag = tag_monoid_of(tag_word, pre, ix);
    const uint32_t path_ix = tag.monoid.path_ix;
    path_ix_out = path_ix;
    em.bx0 = 1e31f; em.by0 = 1e31f; em.bx1 = -1e31f; em.by1 = -1e31f;
    if ((tag.tag_byte & PATH_TAG_SUBPATH_END) != 0u) return true;  // cap marker of a closed subpath: draws nothing (flatten.wgsl:797-807)
    const uint32_t style_ix = tag.monoid.style_ix;
    const uint32_t style_flags = scene[(uint32_t)(cfg.layout.style_base + style_ix)];
    const uint32_t *pd = scene + cfg.layout.path_data_base;
    const Xform transform = read_transform(scene, cfg.layout.transform_base, tag.monoid.trans_ix);
    const CubicPoints pts = read_path_segment(pd, tag, true);
    const float offset = 0.5f * __uint_as_float(scene[cfg.layout.style_base + style_ix + 1u]);
    const float scale = 0.5f * (length(v2(transform.m0 + transform.m3, transform.m1 - transform.m2)) +
                                length(v2(transform.m0 - transform.m3, transform.m1 + transform.m2)));
    // (false for a degenerate segment too: its chord is shorter than the test's lower bound)
    if (!cubic_is_straight(pts.p0, pts.p1, pts.p2, pts.p3, scale, offset)) return false;
    // read_neighboring_segment(ix + 1), flatten.wgsl:810-822 (three times out of four in the word at hand)
    PathTagData ntag = ((ix + 1u) >> 2) == (ix >> 2) ? tag_monoid_of(tag_word, pre, ix + 1u) : compute_tag_monoid(cfg, scene, tag_monoids, ix + 1u);
    CubicPoints npts = read_path_segment(pd, ntag, true);
    bool n_is_closed = (ntag.tag_byte & PATH_TAG_SEG_TYPE) == PATH_TAG_LINETO;
    bool n_is_marker = (ntag.tag_byte & PATH_TAG_SUBPATH_END) != 0u;
    bool do_join = !n_is_marker || n_is_closed;
    // Inputs that are not finite, or so large that the offsets or the transform overflow, go to the heavy code like a line that is not
    // straight (round 6): a NaN makes min / max depend on the order of their operands, and this thread writes a tag's lines -- and the
    // lane that takes its arc the rest -- in another order than flatten.wgsl's one invocation does, so the path's box came out differently
    // (fuzz seeds 4552, 8707, 11797, 11851 with the kernel forced: DESIGN.md 4).  The largest magnitude by the bit patterns: NaN and
    // infinity compare above every number.
    {
        auto mag = [](float f) { return __float_as_uint(f) & 0x7fffffffu; };
        uint32_t m = maxu(maxu(mag(pts.p0.x), mag(pts.p0.y)), maxu(mag(pts.p3.x), mag(pts.p3.y)));
        m = maxu(m, maxu(maxu(mag(npts.p0.x), mag(npts.p0.y)), maxu(mag(npts.p3.x), mag(npts.p3.y))));
        m = maxu(m, maxu(maxu(mag(npts.p1.x), mag(npts.p1.y)), maxu(mag(npts.p2.x), mag(npts.p2.y))));
        m = maxu(m, maxu(maxu(mag(transform.m0), mag(transform.m1)), maxu(mag(transform.m2), mag(transform.m3))));
        m = maxu(m, maxu(maxu(mag(transform.t0), mag(transform.t1)), mag(offset)));
        if (m > __float_as_uint(1.0e15f)) return false;
    }
    vec2 n_tangent = npts.p3 - npts.p0;
    if (!n_is_marker) n_tangent = cubic_start_tangent(npts.p0, npts.p1, npts.p2, npts.p3);
    vec2 tan_start = cubic_start_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
    if (dot(tan_start, tan_start) < TANGENT_THRESH * TANGENT_THRESH) tan_start = v2(TANGENT_THRESH, 0.0f);
    vec2 tan_prev = cubic_end_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
    if (dot(tan_prev, tan_prev) < TANGENT_THRESH * TANGENT_THRESH) tan_prev = v2(TANGENT_THRESH, 0.0f);
    vec2 tan_next = n_tangent;
    if (dot(tan_next, tan_next) < TANGENT_THRESH * TANGENT_THRESH) tan_next = v2(TANGENT_THRESH, 0.0f);
    vec2 n_start = normalize(v2(-tan_start.y, tan_start.x)) * offset;
    vec2 offset_tangent = normalize(tan_prev) * offset;
    vec2 n_prev = v2(-offset_tangent.y, offset_tangent.x);
    vec2 tnn = normalize(tan_next) * offset;
    vec2 n_next = v2(-tnn.y, tnn.x);
    {
        // the two offset lines (flatten_euler's shortcut with two_sided = true)
        const uint32_t line_ix = em.alloc(2u);
        em.write_xf(line_ix, path_ix, pts.p0 + n_start, pts.p3 + n_prev, transform);
        em.write_xf(line_ix + 1u, path_ix, pts.p3 - n_prev, pts.p0 - n_start, transform);
    }
    // the arc of a round join / cap: here if it is certainly one line, else set aside (begin, end, center and what the
    // angle is made of); everything else of draw_join / draw_cap as there
    bool arc = false, arc_is_cap = false;
    vec2 arc0 = v2(0.0f, 0.0f), arc1 = arc0;
    float cr = 0.0f, d = 0.0f;
    if (do_join && (style_flags & STYLE_FLAGS_JOIN_MASK) == STYLE_FLAGS_JOIN_ROUND) {
        const vec2 p0 = pts.p3;
        const vec2 front0 = p0 + n_prev, front1 = p0 + n_next, back0 = p0 - n_next, back1 = p0 - n_prev;
        cr = tan_prev.x * tan_next.y - tan_prev.y * tan_next.x;
        d = dot(tan_prev, tan_next);
        vec2 other0, other1;
        if (cr > 0.0f) { arc0 = back0; arc1 = back1; other0 = front0; other1 = front1; }
        else { arc0 = front0; arc1 = front1; other0 = back0; other1 = back1; }
        arc = !stroke_arc_one_line(em, path_ix, arc0, arc1, p0, cr, d, transform);
        const uint32_t o = em.alloc(1u);
        em.write_xf(o, path_ix, other0, other1, transform);
    } else if (do_join) {
        draw_join<false>(em, path_ix, style_flags, pts.p3, tan_prev, tan_next, n_prev, n_next, transform);
    } else if ((style_flags & STYLE_FLAGS_END_CAP_MASK) == STYLE_FLAGS_CAP_ROUND) {
        arc = arc_is_cap = true;
        arc0 = pts.p3 + n_prev;
        arc1 = pts.p3 - n_prev;
    } else {
        draw_cap<false>(em, path_ix, style_flags & STYLE_FLAGS_END_CAP_MASK, pts.p3, pts.p3 + n_prev, pts.p3 - n_prev, offset_tangent, transform);
    }
    if (arc) {
        const uint32_t k = atomicAdd(&q.count, 1u);  // (< STROKE_ARCS: one per thread and round at most)
        ArcItem it;
        it.path_ix = path_ix; it.trans_ix = tag.monoid.trans_ix; it.flags = arc_is_cap ? ARC_IS_CAP : 0u; it.pad = 0u;
        it.bx = arc0.x; it.by = arc0.y; it.ex = arc1.x; it.ey = arc1.y;
        it.cx = pts.p3.x; it.cy = pts.p3.y; it.cr = cr; it.d = d;
        it.box[0] = em.bx0; it.box[1] = em.by0; it.box[2] = em.bx1; it.box[3] = em.by1;
        q.item[k] = it;
        // (this thread's box travels with the arc: the lane that flattens the arc tests and merges the whole)
        em.bx0 = 1e31f; em.by0 = 1e31f; em.bx1 = -1e31f; em.by1 = -1e31f;
    }
    return true;
}

// ---- the kernels behind k_flatten_light ------------------------------------------------------------------------
// k_flatten_main: two kinds of workgroups in ONE launch (round 3; they were two kernels one after the other: the heavy
// list's long subdivision chains at a few waves per CU, 57-63 us on the road map, and the stroked lines' throughput
// work, 47 us -- neither needs the other's result):
//  * HEAVY workgroups [0, n_heavy_blocks): the lists k_flatten_light left -- curves, strokes and, below
//    FLATTEN_STROKE_KERNEL_MIN_LINES, the stroked lines -- through flatten_tag (Euler-spiral flattener + stroker);
//  * STROKE workgroups behind them (scenes with enough stroked lines to occupy the chip on their own): a thread per line,
//    flatten_tag's stroke branch with flatten_euler reduced to its straight-segment shortcut, round joins that are one
//    line emitted without transcendentals, the other arcs set aside (ArcItem), a line that is not straight after all
//    handed on (heavy_list's fourth section).
// k_flatten_tail: what the stroke workgroups set aside -- the arcs, densely, and the handed-on lines -- by the heavy
// code once more.  Launched only with stroke workgroups.
constexpr uint32_t FLATTEN_LDS_LINES = 2816u;  // 5 words each: 55 KB of staging per workgroup; with the four waves' EulerCoopLds (21 KB; a stroke workgroup: its arc queue, 16 KB) two fit a CU
constexpr uint32_t FLATTEN_STROKE_ROUND_LINES = 1536u;  // what one round of a stroke workgroup (256 stroked lines) emits at most, nearly always
constexpr size_t FLATTEN_ARCS_AT = (sizeof(FlattenShared<FLATTEN_LDS_LINES>) + 15u) & ~(size_t)15u;  // the stroke workgroups' arc queue behind the staging
constexpr size_t FLATTEN_MAIN_LDS = FLATTEN_ARCS_AT + (sizeof(ArcQueue) > 4u * sizeof(EulerCoopLds) ? sizeof(ArcQueue) : 4u * sizeof(EulerCoopLds));  // (the heavy workgroups' EulerCoopLds in the arc queue's place)
static_assert(FLATTEN_MAIN_LDS <= 80u * 1024u && sizeof(FlattenShared<FLATTEN_LDS_LINES>) + 4u * sizeof(EulerCoopLds) <= 80u * 1024u, "two heavy workgroups per CU");


#ifdef VELLO_STROKE_TIMELINE
// measurement build (scripts/stroke_timeline.py): wall-clock stamps (100 MHz) of a stroke workgroup's rounds, thread 0's view
constexpr uint32_t STL_ROUNDS = 16384u;
__device__ uint32_t g_stroke_tl[STL_ROUNDS][8];
__device__ uint32_t g_stroke_tl_n;
#define STL_STAMP(k) do { if (tid == 0u) stl[k] = (uint32_t)wall_clock64(); } while (0)
#else
#define STL_STAMP(k) do { } while (0)
#endif
template <uint32_t CAP>
__device__ __forceinline__ void stroke_workgroup(FlattenShared<CAP> &sh, ArcQueue &arcs, uint32_t block, uint32_t n_blocks, const Config &cfg,
                                 uint32_t n_tags, const uint32_t *__restrict__ scene, const TagMonoid *__restrict__ tag_monoids,
                                 PathBbox *path_bboxes, Control *control, LineSoup *lines, uint32_t *heavy_list, uint32_t min_lines,
                                 uint32_t *arc_items, uint32_t arc_shard_cap) {
    const uint32_t tid = threadIdx.x;
    const uint32_t n_lines_q = control->heavy_count[2];  // final: written by k_flatten_light
    if (n_lines_q < min_lines) return;  // the heavy workgroups take them
    if (block * 256u >= n_lines_q || (control->bump.failed & FAILED_SCENE) != 0u) return;
    if (tid == 0u) {
        sh.count = 0u;
        sh.lds_end = 0xffffffffu;
        arcs.count = 0u;
    }
    __syncthreads();
    Bump *bump = &control->bump;
    const uint32_t lane = tid & 63u;
    // A round is a chain of dependent loads -- list entry, tag word + monoid, points / style / transform, the neighbour's -- and with
    // frames in flight a workgroup walks 7 of them on a road map (profiles/r06_stroke_timeline_d2.txt: the line is 4.4 of a round's
    // 7.1 us): the list entries are requested TWO rounds ahead, the tag words and monoids ONE, from clamped addresses (a load under a
    // branch would wait for everything in flight: DESIGN.md 3.1), so a round starts with its data's addresses in registers.
    const uint32_t stride = n_blocks * 256u, last = n_lines_q - 1u;
    const uint32_t *const list = heavy_list + 2u * (size_t)n_tags;
    uint32_t tag_ix_cur = list[minu(block * 256u + tid, last)];
    uint32_t tag_ix_next = list[minu(block * 256u + tid + stride, last)];
    uint32_t word_cur = scene[cfg.layout.path_tag_base + (tag_ix_cur >> 2)];
    TagMonoid pre_cur = tag_monoids[tag_ix_cur >> 2];
#pragma unroll 1
    for (uint32_t base = block * 256u; base < n_lines_q; base += stride) {
        // (what the rounds after this one will want; beyond the list: its last entry again)
        const uint32_t tag_ix_after = list[minu(base + tid + 2u * stride, last)];
        const uint32_t word_next = scene[cfg.layout.path_tag_base + (tag_ix_next >> 2)];
        const TagMonoid pre_next = tag_monoids[tag_ix_next >> 2];
#ifdef VELLO_STROKE_TIMELINE
        uint32_t stl[8] = {};
#endif
        STL_STAMP(0);
        Emitter em;
        em.lines = lines;
        em.lines_size = cfg.lines_size;
        em.bump = bump;
        em.bind(sh);
        uint32_t key = 0xffffffffu;
        float x0 = 1e31f, y0 = 1e31f, x1 = -1e31f, y1 = -1e31f;
        const uint32_t e = base + tid;
        bool hand_on = false;
        const uint32_t tag_ix = tag_ix_cur;
        if (e < n_lines_q) {
            hand_on = !flatten_stroked_line(em, arcs, cfg, scene, tag_monoids, tag_ix, word_cur, pre_cur, key);
            if (hand_on) key = 0xffffffffu;
            else if (em.bx1 > em.bx0 || em.by1 > em.by0) {
                x0 = em.bx0; y0 = em.by0; x1 = em.bx1; y1 = em.by1;
            }
        }
        // (rare: one atomic each is fine.  A list of its own: the heavy workgroups of this launch read the lengths of the
        // other three while this one grows)
        STL_STAMP(1);
        if (hand_on) heavy_list[3u * n_tags + atomicAdd(&control->heavy_count[3], 1u)] = tag_ix;
        // The round's arcs go to the frame's list and -- when another round would no longer fit the staging area, or there is none --
        // the staged lines to the soup: the two reservations as ONE instruction of two lanes (they were two round trips in a row;
        // the soup's counter is one address for every workgroup of the launch: keep staging while another round still fits).
        __syncthreads();  // (sh.count, arcs.count: every emit of the round is behind this)
        STL_STAMP(2);
        {
            const uint32_t n_arcs = arcs.count;
            const uint32_t shard = block % FLATTEN_ARC_SHARDS;
            const bool last_round = base + n_blocks * 256u >= n_lines_q;
            const bool do_flush = last_round || sh.count + FLATTEN_STROKE_ROUND_LINES > CAP;
            const uint32_t n_lds = do_flush ? minu(sh.count, sh.lds_end) : 0u;
            // (the paths' boxes between the reservations' request and their answer: the box atomics need neither -- round 6)
            uint32_t got = 0u;
            if (tid < 2u) {
                uint32_t *const counter = tid == 0u ? &control->arc_count[shard] : &bump->lines;
                const uint32_t n = tid == 0u ? n_arcs : n_lds;
                got = n ? atomicAdd(counter, n) : 0u;
            }
            wave_bbox_update(path_bboxes, cfg.layout.n_paths, key, x0, y0, x1, y1, (int)lane);
            if (tid == 0u) arcs.base = got;
            if (tid == 1u) sh.base = got;
#ifdef VELLO_STROKE_TIMELINE
            __builtin_amdgcn_s_waitcnt(0);
#endif
            STL_STAMP(3);
            __syncthreads();
            STL_STAMP(4);
            // (a shard holds <= 256 arcs per round of each of its workgroups: arc_shard_cap is sized for that)
            if (tid < n_arcs && arcs.base + tid < arc_shard_cap) reinterpret_cast<ArcItem *>(arc_items)[shard * arc_shard_cap + arcs.base + tid] = arcs.item[tid];
            if (do_flush) copy_staged_lines(sh, lines, cfg.lines_size, sh.base, n_lds, tid);
#ifdef VELLO_STROKE_TIMELINE
            __builtin_amdgcn_s_waitcnt(0);
#endif
            STL_STAMP(5);
            __syncthreads();
            STL_STAMP(6);
#ifdef VELLO_STROKE_TIMELINE
            if (tid == 0u) {
                const uint32_t slot = atomicAdd(&g_stroke_tl_n, 1u);
                stl[7] = n_lds | (n_arcs << 16);
                if (slot < STL_ROUNDS)
                    for (uint32_t k = 0; k < 8u; k++) g_stroke_tl[slot][k] = stl[k];
            }
#endif
            if (tid == 0u) {
                arcs.count = 0u;
                if (do_flush) {
                    sh.count = 0u;
                    sh.lds_end = 0xffffffffu;
                }
            }
            __syncthreads();  // (the next round appends to both)
        }
        tag_ix_cur = tag_ix_next;
        tag_ix_next = tag_ix_after;
        word_cur = word_next;
        pre_cur = pre_next;
    }
}

// The heavy code over its list.  LISTS & HEAVY_FIRST: [curves | strokes | stroked lines below the threshold] as k_flatten_light
// left them; LISTS & HEAVY_SET_ASIDE: [handed-on lines | arcs] as stroke workgroups -- of an EARLIER launch -- left them.
constexpr uint32_t HEAVY_FIRST = 1u, HEAVY_SET_ASIDE = 2u;
#ifndef VK_FL_LPW_DIV
#define VK_FL_LPW_DIV 1024u          // a long list is spread over this many waves ...
#define VK_FL_LPW_DIV_STROKES 3072u  // ... a list with many stroked curves over this many (sweep constants)
#endif
template <uint32_t LISTS, bool COOP>
__device__ __forceinline__ void heavy_workgroups(FlattenShared<FLATTEN_LDS_LINES> &sh, EulerCoopLds *coop, uint32_t block, uint32_t n_blocks, const Config &cfg, uint32_t n_tags,
                                                 const uint32_t *__restrict__ scene, const TagMonoid *__restrict__ tag_monoids, PathBbox *path_bboxes,
                                                 Control *control, LineSoup *lines, const uint32_t *__restrict__ heavy_list,
                                                 uint32_t stroke_kernel_min_lines, const uint32_t *__restrict__ arc_items, uint32_t arc_shard_cap) {
    constexpr bool FIRST = (LISTS & HEAVY_FIRST) != 0u, ASIDE = (LISTS & HEAVY_SET_ASIDE) != 0u;
    const uint32_t tid = threadIdx.x;
    // final counts: written by the previous kernel on this stream
    const uint32_t n_curves = FIRST ? control->heavy_count[0] : 0u;
    const uint32_t n_strokes = FIRST ? control->heavy_count[1] : 0u;
    const uint32_t n_handed = ASIDE ? control->heavy_count[3] : 0u;
    const uint32_t n_lines_q = FIRST && control->heavy_count[2] < stroke_kernel_min_lines ? control->heavy_count[2] : 0u;  // else the stroke workgroups' work
    // the arcs the stroke workgroups set aside are the list's last section: shard s holds arc_incl[s] - arc_excl of them
    static_assert(FLATTEN_ARC_SHARDS == 64u, "a lane per shard");
    const uint32_t arc_n = ASIDE ? minu(control->arc_count[tid & 63u], arc_shard_cap) : 0u;
    const uint32_t arc_incl = wave_incl_scan_u32(arc_n, (int)(tid & 63u));
    const uint32_t n_arcs = wave_read(arc_incl, 63u);
    const uint32_t n_tag_entries = n_curves + n_strokes + n_handed + n_lines_q;
    const uint32_t n_heavy = n_tag_entries + n_arcs;
    // Lanes of a wave walk DIFFERENT subdivision trees, so a wave executes the union of its lanes' loops: as long as the
    // launch has more waves than the list has entries to fill them, every wave takes only as many entries as it must
    // (a 900-curve SVG gets a wave per curve on 900 of the chip's 2048 wave slots instead of 64 curves in each of 15
    // waves on four CUs); a long list fills the waves completely and strides.
    const uint32_t n_waves = n_blocks * 4u;
    // list entries per wave.  Up to 4 096 entries: ONE -- a wave per entry, two rounds of the chip's 2 048 wave slots at
    // most, each wave as long as its own curve (a second entry in a wave makes the wave the union of two subdivision
    // loops: tiger flatten 190 -> 146 us against two entries per wave).  Beyond: pack -- the launch is the sum of its
    // waves, and denser waves are fewer of them (mmark-50k 334 -> 294 us with 49 entries per wave against 25; the road map
    // pays 6 us one frame at a time and gains 3 % with frames in flight).  Round 3, measured over 1 / 2 / 4 K divisors.
    // Round 5 (the Euler flattener is wave-cooperative now: what a wave pays per entry is the union of its lanes' SUBDIVISION walks,
    // the lines are spread over all lanes whoever owns them): a list with many stroked curves (two offset curves per walk, a dozen
    // turns, wildly different from lane to lane) is spread over three times as many waves -- mmark-50k flatten 204 -> 182 us -- while
    // fills' curves, stroked lines and arcs stay dense (the road map's blobs and arcs: 136 us dense, 170 at a third of the density;
    // profiles/r05_ab_flatten_coop.txt).
    // (n_strokes counts the cap markers of open subpaths too -- a QUADTO tag each, vello_encoding/src/path.rs -- hence "most of the list")
    const uint32_t lpw_div = n_strokes * 2u > n_heavy + n_curves * 2u ? VK_FL_LPW_DIV_STROKES : VK_FL_LPW_DIV;
    const uint32_t lpw = n_heavy <= 4096u ? 1u : minu(maxu((n_heavy + lpw_div - 1u) / lpw_div, 1u), 64u);
    if (block * 4u * lpw >= n_heavy || (control->bump.failed & FAILED_SCENE) != 0u) return;
    if (tid == 0u) {
        sh.count = 0u;
        sh.lds_end = 0xffffffffu;
    }
    __syncthreads();
    Bump *bump = &control->bump;
    const uint32_t lane = tid & 63u, wave = block * 4u + (tid >> 6);
    flp_start();
    // no indirect dispatch in HIP: a fixed grid strides over the list
#pragma unroll 1
    for (uint32_t base = 0u; base + block * 4u * lpw < n_heavy; base += n_waves * lpw) {
        Emitter em;
        em.lines = lines;
        em.lines_size = cfg.lines_size;
        em.bump = bump;
        em.bind(sh);
        uint32_t key = 0xffffffffu;
        float x0 = 1e31f, y0 = 1e31f, x1 = -1e31f, y1 = -1e31f;
        const uint32_t e = base + wave * lpw + lane;
        // which arc of which shard entry e is (if it is one): the shards' inclusive counts sit one per lane; the search is
        // taken by all lanes together (shuffles)
        uint32_t arc_shard = 0u, arc_local = 0u;
        if (ASIDE && n_arcs != 0u) {
            const uint32_t a = e >= n_tag_entries ? e - n_tag_entries : 0u;
#pragma unroll
            for (uint32_t step = 32u; step >= 1u; step >>= 1)
                if (wave_shfl(arc_incl, arc_shard + step - 1u) <= a) arc_shard += step;
            arc_shard = minu(arc_shard, 63u);
            arc_local = a - wave_shfl(arc_incl - arc_n, arc_shard);
        }
        // (every lane of the wave goes through flatten_tag_coop: the lanes without an entry flatten the others' lines)
        const bool has_tag = lane < lpw && e < n_tag_entries;
        uint32_t tag_ix = 0u;
        if (has_tag) {
            const uint32_t e1 = e - n_curves, e3 = e1 - n_strokes, e2 = e3 - n_handed;
            tag_ix = e < n_curves     ? heavy_list[e]
                     : e1 < n_strokes ? heavy_list[n_tags + e1]
                     : e3 < n_handed  ? heavy_list[3u * n_tags + e3]
                                      : heavy_list[2u * n_tags + e2];
        }
        // (COOP is the host's choice for the scene, engine.hip Frame::flatten_coop: the kernels of the cooperative walk, with the fp64
        // routines out of line, or round 4's, every lane on its own with the routines inline -- flatten_walk.inc's head says why)
        uint32_t tag_key = 0u;
        if (COOP) tag_key = flatten_tag_coop(em, coop[tid >> 6], has_tag, cfg, scene, tag_monoids, tag_ix, lane, lpw <= 4u);
        else if (has_tag) tag_key = inl::flatten_tag(em, cfg, scene, tag_monoids, path_bboxes, tag_ix);
        if (has_tag) {
            key = tag_key;
            if (em.bx1 > em.bx0 || em.by1 > em.by0) {
                x0 = em.bx0; y0 = em.by0; x1 = em.bx1; y1 = em.by1;
            }
        } else if (ASIDE && lane < lpw && e < n_heavy) {
            // an arc a stroke workgroup set aside: flatten_arc as draw_join / draw_cap call it, then the box of the thread
            // that met it -- its other lines' and this arc's -- tested and merged as one (flatten.wgsl:916-921)
            const ArcItem it = reinterpret_cast<const ArcItem *>(arc_items)[arc_shard * arc_shard_cap + arc_local];
            em.bx0 = 1e31f; em.by0 = 1e31f; em.bx1 = -1e31f; em.by1 = -1e31f;
            const Xform t = read_transform(scene, cfg.layout.transform_base, it.trans_ix);
            if (COOP) {
                const float angle = (it.flags & ARC_IS_CAP) != 0u ? 3.1415927f : fabsf(fl_atan2(it.cr, it.d));
                outl::flatten_arc(em, it.path_ix, v2(it.bx, it.by), v2(it.ex, it.ey), v2(it.cx, it.cy), angle, t);
            } else {
                const float angle = (it.flags & ARC_IS_CAP) != 0u ? 3.1415927f : fabsf(atan2_cr(it.cr, it.d));
                inl::flatten_arc(em, it.path_ix, v2(it.bx, it.by), v2(it.ex, it.ey), v2(it.cx, it.cy), angle, t);
            }
            key = it.path_ix;
            const float mx0 = minf(em.bx0, it.box[0]), my0 = minf(em.by0, it.box[1]), mx1 = maxf(em.bx1, it.box[2]), my1 = maxf(em.by1, it.box[3]);
            if (mx1 > mx0 || my1 > my0) {
                x0 = mx0; y0 = my0; x1 = mx1; y1 = my1;
            }
        }
        // list entries of one source workgroup keep tag order, so equal path keys still come in runs
        flp_mark(FLP_OTHER);
        wave_bbox_update(path_bboxes, cfg.layout.n_paths, key, x0, y0, x1, y1, (int)lane);
        flush_staged_lines(sh, bump, lines, cfg.lines_size, tid);
        flp_mark(FLP_BBOX_FLUSH);
    }
    flp_store();
}

template <bool COOP>
__global__ void __launch_bounds__(256, 2) k_flatten_main(Config cfg, uint32_t n_tags, const uint32_t *__restrict__ scene,
                                                         const TagMonoid *__restrict__ tag_monoids, PathBbox *path_bboxes,
                                                         Control *control, LineSoup *lines, uint32_t *heavy_list,
                                                         uint32_t stroke_kernel_min_lines, uint32_t *arc_items, uint32_t arc_shard_cap,
                                                         uint32_t n_heavy_blocks) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[FLATTEN_MAIN_LDS];
    if (blockIdx.x < n_heavy_blocks) {
        // (the heavy workgroups keep their four waves' EulerCoopLds where the stroke workgroups keep their arc queue)
        heavy_workgroups<HEAVY_FIRST, COOP>(*reinterpret_cast<FlattenShared<FLATTEN_LDS_LINES> *>(smem), reinterpret_cast<EulerCoopLds *>(smem + FLATTEN_ARCS_AT),
                                      blockIdx.x, n_heavy_blocks, cfg, n_tags, scene, tag_monoids,
                                      path_bboxes, control, lines, heavy_list, stroke_kernel_min_lines, arc_items, arc_shard_cap);
    } else {
        stroke_workgroup(*reinterpret_cast<FlattenShared<FLATTEN_LDS_LINES> *>(smem), *reinterpret_cast<ArcQueue *>(smem + FLATTEN_ARCS_AT),
                         blockIdx.x - n_heavy_blocks, gridDim.x - n_heavy_blocks, cfg, n_tags, scene, tag_monoids, path_bboxes, control, lines,
                         heavy_list, stroke_kernel_min_lines, arc_items, arc_shard_cap);
    }
}

__global__ void __launch_bounds__(256, 2) k_flatten_tail(Config cfg, uint32_t n_tags, const uint32_t *__restrict__ scene,
                                                         const TagMonoid *__restrict__ tag_monoids, PathBbox *path_bboxes,
                                                         Control *control, LineSoup *lines, const uint32_t *__restrict__ heavy_list,
                                                         const uint32_t *__restrict__ arc_items, uint32_t arc_shard_cap) {
    __shared__ FlattenShared<FLATTEN_LDS_LINES> sh;
    // (what the stroke workgroups set aside -- arcs, and lines that were not straight after all: every lane on its own)
    heavy_workgroups<HEAVY_SET_ASIDE, false>(sh, nullptr, blockIdx.x, gridDim.x, cfg, n_tags, scene, tag_monoids, path_bboxes, control, lines, heavy_list, 0u, arc_items,
                                      arc_shard_cap);
}

// ... and the same work as two launches one after the other, for frames in flight: what k_flatten_main gains by running the
// two kinds side by side only matters to a frame that has the chip to itself, and its stroke workgroups hold the 256
// registers and 76 KB its heavy ones need -- room the kernels of the other frames would use (76 registers, 47 KB here:
// a road map renders 2.5 % faster with frames in flight this way, 6 % slower one at a time; round 3, same-box A/B).
__global__ void __launch_bounds__(256) k_flatten_strokes(Config cfg, uint32_t n_tags, const uint32_t *__restrict__ scene,
                                                        const TagMonoid *__restrict__ tag_monoids, PathBbox *path_bboxes, Control *control,
                                                        LineSoup *lines, uint32_t *heavy_list, uint32_t min_lines, uint32_t *arc_items,
                                                        uint32_t arc_shard_cap) {
    __shared__ FlattenShared<FLATTEN_STROKE_ROUND_LINES> sh;  // 30 KB of staging + 16 KB of arcs: three workgroups per CU
    __shared__ ArcQueue arcs;
    stroke_workgroup(sh, arcs, blockIdx.x, gridDim.x, cfg, n_tags, scene, tag_monoids, path_bboxes, control, lines, heavy_list, min_lines, arc_items,
                     arc_shard_cap);
}

template <bool COOP>
__global__ void __launch_bounds__(256, 2) k_flatten_heavy(Config cfg, uint32_t n_tags, const uint32_t *__restrict__ scene,
                                                          const TagMonoid *__restrict__ tag_monoids, PathBbox *path_bboxes,
                                                          Control *control, LineSoup *lines, const uint32_t *__restrict__ heavy_list,
                                                          uint32_t stroke_kernel_min_lines, const uint32_t *__restrict__ arc_items,
                                                          uint32_t arc_shard_cap) {
    __shared__ FlattenShared<FLATTEN_LDS_LINES> sh;
    __shared__ EulerCoopLds coop[COOP ? 4 : 1];
    heavy_workgroups<HEAVY_FIRST | HEAVY_SET_ASIDE, COOP>(sh, coop, blockIdx.x, gridDim.x, cfg, n_tags, scene, tag_monoids, path_bboxes, control, lines, heavy_list,
                                                    stroke_kernel_min_lines, arc_items, arc_shard_cap);
}

// ---- small scenes: consecutive stages as ONE launch ------------------------------------------------------------------
// A frame is a chain of a dozen dependent launches, and a launch boundary on this machine is ~4 us of nothing plus the ramp of
// the next grid: for the 48-line circle that is the whole frame, for the Tiger a fifth of it.  k_front runs the workgroups of
// several stages -- the same bodies, block by block -- as turns of a few persistent workgroups with a grid barrier between the
// stages: [zero fill of the control block | pathtag scan | flatten's light pass + draw scan], and [binning | tile_alloc]; for
// a scene of a few dozen segments the heavy list joins them and everything up to tile_alloc is one workgroup's work, the
// barriers plain __syncthreads.  The barrier: one release-add per workgroup on a counter that only ever grows (the host
// knows its value at the launch: `sync_base`), a spin until all have arrived, an acquire.  At most FRONT_MAX_WG workgroups,
// so that every cooperative launch that can be in flight at once (frames in flight, other contexts) is resident as a whole.
// Reference: vello/src/render.rs:250-436 (the dispatch chain these stages are).
struct FrontArgs {
    uint32_t stages;
    uint32_t n_tag_words, n_scene_words, n_tags;
    uint32_t n_pathtag_blocks, n_draw_blocks, n_light_blocks, n_binning_blocks, n_tile_alloc_blocks;
    uint32_t arc_shard_cap;
    uint32_t zero_vec16;  // 16-byte words of the lane's zero region (Control + look-back states)
    uint32_t sync_base;   // *sync when the launch begins
    uint32_t *sync;
    const uint32_t *scene;
    Control *control;
    unsigned long long *pathtag_state, *draw_state;
    TagMonoid *tag_monoids;
    PathBbox *path_bboxes;
    LineSoup *lines;
    uint32_t *heavy_list, *arc_items;
    DrawMonoid *draw_monoids;
    uint32_t *info_bin_data;
    Clip *clip_inp;
    Bbox4 *clip_bboxes, *draw_bboxes;
    BinHeader *bin_headers;
    Path *paths;
    Tile *tiles;
};

// All workgroups of the launch have finished the stage before / may start the next one.  Returns false when the wait gave up (or
// had given up before: `tripped`).  A workgroup that gave up keeps ADDING at every later barrier -- the counter must end at
// sync_base + (barriers of the launch) x (workgroups), or every later launch on the lane would wait for the difference -- but no
// longer waits, and k_front skips its remaining stage bodies: their inputs are whatever the workgroups it did not wait for had
// written by then, or the previous frame's.
__device__ __forceinline__ bool front_barrier(uint32_t *sync, uint32_t target, bool tripped, uint32_t *sh_tripped) {
    __syncthreads();  // (the workgroup's stores are out)
#ifndef VELLO_SIMT_EMU  // (the emulator runs workgroups one after the other: a launch there is ONE workgroup)
    if (gridDim.x != 1u) {
        if (threadIdx.x == 0u) {
            __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0u;
            while (!tripped && (int32_t)(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                // (as the look-back's: a launch that is not resident as a whole -- it always is, FRONT_MAX_WG -- must fail, not hang)
                if (++spins > SPIN_LIMIT) {
                    tripped = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            *sh_tripped = tripped ? 1u : 0u;
        }
        __syncthreads();
        tripped = *sh_tripped != 0u;
    }
#else
    (void)sync; (void)target; (void)sh_tripped;
#endif
    return !tripped;
}

template <bool HEAVY>
__global__ void __launch_bounds__(256) k_front(Config cfg, FrontArgs a) {
    const uint32_t tid = threadIdx.x, wg = blockIdx.x, n_wg = gridDim.x;
    uint32_t target = a.sync_base;
    uint32_t todo = a.stages;
    __shared__ uint32_t sh_tripped;
    bool ok = true;  // (workgroup-uniform) no barrier of this launch has given up so far
    // (a barrier behind every stage but the launch's last)
#define FRONT_STAGE_END(bit)                                   \
    do {                                                       \
        todo &= ~(bit);                                        \
        if (todo != 0u) ok = front_barrier(a.sync, target += n_wg, !ok, &sh_tripped); \
    } while (0)
    if (a.stages & FRONT_ZERO) {  // render.rs:313 clears `bump`; here: Control and both look-back states
        uint4 *z = reinterpret_cast<uint4 *>(a.control);
        for (uint32_t i = wg * 256u + tid; i < a.zero_vec16; i += n_wg * 256u) z[i] = make_uint4(0u, 0u, 0u, 0u);
        FRONT_STAGE_END(FRONT_ZERO);
    }
    if (a.stages & FRONT_PATHTAG) {
        for (uint32_t b = wg; ok && b < a.n_pathtag_blocks; b += n_wg) {
            pathtag_scan_workgroup(cfg, b, a.n_pathtag_blocks, a.n_tag_words, a.n_scene_words, a.scene, a.control, a.pathtag_state, a.tag_monoids,
                                   a.path_bboxes);
            __syncthreads();  // (the next turn writes the LDS this one read)
        }
        FRONT_STAGE_END(FRONT_PATHTAG);
    }
    if (a.stages & FRONT_LIGHT) {
        for (uint32_t b = wg; ok && b < a.n_draw_blocks + a.n_light_blocks; b += n_wg) {
            if (b < a.n_draw_blocks)
                draw_scan_workgroup(cfg, a.scene, a.control, a.draw_state, a.path_bboxes, a.draw_monoids, a.info_bin_data, a.clip_inp);
            else
                flatten_light_workgroup(cfg, b - a.n_draw_blocks, a.n_tags, a.scene, a.tag_monoids, a.path_bboxes, a.control, a.lines, a.heavy_list);
            __syncthreads();
        }
        FRONT_STAGE_END(FRONT_LIGHT);
    }
    if constexpr (HEAVY) {
        if (a.stages & FRONT_HEAVY) {
            // (no stroke workgroups: every stroked line is the heavy list's; the workgroups stride over the list themselves)
            __shared__ __attribute__((aligned(16))) unsigned char smem[FLATTEN_MAIN_LDS];
            if (ok)
            heavy_workgroups<HEAVY_FIRST, true>(*reinterpret_cast<FlattenShared<FLATTEN_LDS_LINES> *>(smem),
                                                reinterpret_cast<EulerCoopLds *>(smem + FLATTEN_ARCS_AT), wg, n_wg, cfg, a.n_tags, a.scene, a.tag_monoids,
                                                a.path_bboxes, a.control, a.lines, a.heavy_list, 0xffffffffu, a.arc_items, a.arc_shard_cap);
            FRONT_STAGE_END(FRONT_HEAVY);
        }
    }
    if (a.stages & FRONT_BINNING) {
        for (uint32_t b = wg; ok && b < a.n_binning_blocks; b += n_wg) {
            binning_workgroup(cfg, b, a.draw_monoids, a.path_bboxes, a.clip_bboxes, a.draw_bboxes, &a.control->bump, a.info_bin_data, a.bin_headers);
            __syncthreads();
        }
        FRONT_STAGE_END(FRONT_BINNING);
    }
    if (a.stages & FRONT_TILE_ALLOC) {
        for (uint32_t b = wg; ok && b < a.n_tile_alloc_blocks; b += n_wg) {
            tile_alloc_workgroup(cfg, b, a.scene, a.draw_bboxes, &a.control->bump, a.paths, a.tiles);
            __syncthreads();
        }
    }
    // (raised at the END of the launch: a flag set at the barrier behind FRONT_ZERO could be cleared again by the zero fill of a
    // workgroup that had not got that far)
    if (!ok && tid == 0u) atomicOr(&a.control->bump.failed, FAILED_INTERNAL);
#undef FRONT_STAGE_END
}

#ifdef VELLO_STROKE_TIMELINE
}  // namespace vk
// measurement build only: the stamps of the stroke workgroups' rounds since the last call, read and cleared (scripts/stroke_timeline.py)
extern "C" int vello_stroke_timeline_read(uint32_t *out, uint32_t *n_out) {
    uint32_t zero = 0u;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(n_out, HIP_SYMBOL(vk::g_stroke_tl_n), 4);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(vk::g_stroke_tl), sizeof(uint32_t) * 8u * vk::STL_ROUNDS);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(vk::g_stroke_tl_n), &zero, 4);
    return (int)e;
}
namespace vk {
#endif
#ifdef VELLO_FLATTEN_PROF
}  // namespace vk
// measurement build only: the counters of g_flatten_prof, read and cleared (scripts/flatten_prof.py binds it with ctypes)
extern "C" int vello_flatten_prof_read(unsigned long long *out) {
    unsigned long long zero[2 * vk::FLP_PHASES + vk::FLC_COUNTS + 2] = {};
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(vk::g_flatten_prof), sizeof(zero));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(vk::g_flatten_prof), zero, sizeof(zero));
    return (int)e;
}
namespace vk {
#endif

constexpr uint32_t FRONT_MAX_WG = 16u;
uint32_t launch_front(const Frame &f, hipStream_t s, uint32_t stages, bool with_draw_scan, uint32_t sync_base) {
    FrontArgs a{};
    a.stages = stages;
    a.n_tag_words = f.n_tag_words;
    a.n_scene_words = f.n_scene_words;
    a.n_tags = f.n_tag_words * 4u;
    a.n_pathtag_blocks = (f.n_tag_words + PATHTAG_PART_WORDS - 1u) / PATHTAG_PART_WORDS;
    if (a.n_pathtag_blocks == 0u) a.n_pathtag_blocks = 1u;
    a.n_draw_blocks = with_draw_scan ? (f.cfg.layout.n_draw_objects + DRAW_PART - 1u) / DRAW_PART : 0u;
    a.n_light_blocks = (a.n_tags + FLATTEN_BLOCK_TAGS - 1u) / FLATTEN_BLOCK_TAGS;
    a.n_binning_blocks = (f.cfg.layout.n_draw_objects + 255u) / 256u;
    a.n_tile_alloc_blocks = (f.cfg.layout.n_paths + 255u) / 256u;
    a.arc_shard_cap = flatten_arc_shard_cap(flatten_n_seg_max(f), true);
    a.zero_vec16 = f.zero_bytes / 16u;
    a.sync_base = sync_base;
    a.sync = f.front_sync;
    a.scene = f.scene;
    a.control = f.control;
    a.pathtag_state = f.pathtag_state;
    a.draw_state = f.draw_state;
    a.tag_monoids = f.tag_monoids;
    a.path_bboxes = f.path_bboxes;
    a.lines = f.lines;
    a.heavy_list = f.heavy_list;
    a.arc_items = f.arc_items;
    a.draw_monoids = f.draw_monoids;
    a.info_bin_data = f.info_bin_data;
    a.clip_inp = f.clip_inp;
    a.clip_bboxes = f.clip_bboxes;
    a.draw_bboxes = f.draw_bboxes;
    a.bin_headers = f.bin_headers;
    a.paths = f.paths;
    a.tiles = f.tiles;
    // as many workgroups as the widest of the launch's stages has blocks, FRONT_MAX_WG at most; one when the heavy list is aboard
    uint32_t n_wg = 1u;
#ifndef VELLO_SIMT_EMU
    if ((stages & FRONT_HEAVY) == 0u) {
        if (stages & FRONT_PATHTAG) n_wg = n_wg > a.n_pathtag_blocks ? n_wg : a.n_pathtag_blocks;
        if (stages & FRONT_LIGHT) n_wg = n_wg > a.n_draw_blocks + a.n_light_blocks ? n_wg : a.n_draw_blocks + a.n_light_blocks;
        if (stages & FRONT_BINNING) n_wg = n_wg > a.n_binning_blocks ? n_wg : a.n_binning_blocks;
        if (stages & FRONT_TILE_ALLOC) n_wg = n_wg > a.n_tile_alloc_blocks ? n_wg : a.n_tile_alloc_blocks;
        if (n_wg > FRONT_MAX_WG) n_wg = FRONT_MAX_WG;
    }
#endif
    if (stages & FRONT_HEAVY) hipLaunchKernelGGL(k_front<true>, dim3(n_wg), dim3(256), 0, s, f.cfg, a);
    else hipLaunchKernelGGL(k_front<false>, dim3(n_wg), dim3(256), 0, s, f.cfg, a);
    uint32_t n_stages = 0u;
    for (uint32_t b = stages; b != 0u; b &= b - 1u) n_stages++;
    return n_wg == 1u ? 0u : n_wg * (n_stages - 1u);
}

void launch_flatten(const Frame &f, hipStream_t s, hipEvent_t *mid, bool with_draw_scan, bool light_done) {
    uint32_t n_tags = f.n_tag_words * 4u;
    uint32_t grid = (n_tags + FLATTEN_BLOCK_TAGS - 1u) / FLATTEN_BLOCK_TAGS;
    if (grid == 0) {
        if (with_draw_scan && !light_done) launch_draw_scan(f, s);
        // (the stage's in-between events are recorded on every way out: vello_hip_get_kernel_ms reads all of them)
        if (mid) {
            (void)hipEventRecord(mid[0], s);
            (void)hipEventRecord(mid[1], s);
        }
        return;
    }
    const uint32_t grid_draw = with_draw_scan ? (f.cfg.layout.n_draw_objects + DRAW_PART - 1u) / DRAW_PART : 0u;
    if (!light_done)
        hipLaunchKernelGGL(k_flatten_light, dim3(grid + grid_draw), dim3(256), 0, s, f.cfg, n_tags, f.scene, f.tag_monoids, f.path_bboxes, f.control,
                           f.lines, f.heavy_list, grid_draw, f.draw_state, f.draw_monoids, f.info_bin_data, f.clip_inp);
    if (mid) (void)hipEventRecord(mid[0], s);
    // enough workgroups for a wave per list entry on small scenes and for one round per workgroup on large ones
    // (workgroups beyond the list exit at once)
    // (a segment owns at least one word of path data, so the path-data stream bounds the list even though the tag stream
    // is padded to 4096 tags)
    const uint32_t n_seg_max = flatten_n_seg_max(f);
    uint32_t grid_heavy = (n_seg_max + 3u) / 4u;
#ifndef VK_FH_GRID_IN_FLIGHT
#define VK_FH_GRID_IN_FLIGHT 2048u  // (sweep constant: the heavy workgroups' cap with frames in flight)
#endif
    const uint32_t heavy_cap = f.flatten_side_by_side ? 2048u : VK_FH_GRID_IN_FLIGHT;
    if (grid_heavy > heavy_cap) grid_heavy = heavy_cap;
    if (grid_heavy < 4u) grid_heavy = 4u;
    // (stroke workgroups leave at once when the scene has too few stroked lines for workgroups of their own; they are not
    // launched at all once a finished frame of the scene has shown that)
    const uint32_t grid_strokes = f.launch_stroke_kernel ? flatten_strokes_grid(n_seg_max, f.flatten_side_by_side) : 0u;
    const uint32_t arc_shard_cap = flatten_arc_shard_cap(n_seg_max, f.flatten_side_by_side);
    const uint32_t min_lines = f.launch_stroke_kernel ? f.stroke_kernel_min_lines : 0xffffffffu;  // (no stroke workgroups: every line is the heavy ones')
    if (f.flatten_side_by_side) {
        if (f.flatten_coop)
            hipLaunchKernelGGL(k_flatten_main<true>, dim3(grid_heavy + grid_strokes), dim3(256), 0, s, f.cfg, n_tags, f.scene, f.tag_monoids, f.path_bboxes,
                               f.control, f.lines, f.heavy_list, min_lines, f.arc_items, arc_shard_cap, grid_heavy);
        else
            hipLaunchKernelGGL(k_flatten_main<false>, dim3(grid_heavy + grid_strokes), dim3(256), 0, s, f.cfg, n_tags, f.scene, f.tag_monoids, f.path_bboxes,
                               f.control, f.lines, f.heavy_list, min_lines, f.arc_items, arc_shard_cap, grid_heavy);
        if (mid) (void)hipEventRecord(mid[1], s);
        // what the stroke workgroups set aside (arcs: a few per cent of the lines; handed-on lines: nearly none)
        if (grid_strokes != 0u) {
            uint32_t grid_tail = (n_seg_max / 16u + 3u) / 4u;
            if (grid_tail > 1024u) grid_tail = 1024u;
            if (grid_tail < 4u) grid_tail = 4u;
            hipLaunchKernelGGL(k_flatten_tail, dim3(grid_tail), dim3(256), 0, s, f.cfg, n_tags, f.scene, f.tag_monoids, f.path_bboxes, f.control,
                               f.lines, f.heavy_list, f.arc_items, arc_shard_cap);
        }
    } else {
        if (grid_strokes != 0u)
            hipLaunchKernelGGL(k_flatten_strokes, dim3(grid_strokes), dim3(256), 0, s, f.cfg, n_tags, f.scene, f.tag_monoids, f.path_bboxes,
                               f.control, f.lines, f.heavy_list, f.stroke_kernel_min_lines, f.arc_items, arc_shard_cap);
        if (mid) (void)hipEventRecord(mid[1], s);
        if (f.flatten_coop)
            hipLaunchKernelGGL(k_flatten_heavy<true>, dim3(grid_heavy), dim3(256), 0, s, f.cfg, n_tags, f.scene, f.tag_monoids, f.path_bboxes, f.control,
                               f.lines, f.heavy_list, min_lines, f.arc_items, arc_shard_cap);
        else
            hipLaunchKernelGGL(k_flatten_heavy<false>, dim3(grid_heavy), dim3(256), 0, s, f.cfg, n_tags, f.scene, f.tag_monoids, f.path_bboxes, f.control,
                               f.lines, f.heavy_list, min_lines, f.arc_items, arc_shard_cap);
    }
}

}  // namespace vk
