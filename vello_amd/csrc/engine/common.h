// Shared definitions for the gfx950 kernels: reference byte layouts,
// numeric rules, and wave64 / workgroup primitives.
//
// Numeric rules (DESIGN.md "numerics"):
//  * compiled with -ffp-contract=off -fno-fast-math: the tile/pixel DDA in path_count, path_tiling
//    and fine recomputes floor(a*i+b) three times and must agree bit for bit;
//  * fma only where flatten.wgsl:668-672 spells fma();
//  * f32 transcendentals = an fp64 evaluation (ocml, or fp64_math.h for sin / cos) rounded once to f32,
//    which equals the correctly rounded result up to ~2^-28 per call and is what the CPU oracle computes
//    with libm;
//  * WGSL u32(f32)/i32(f32) are saturating, round() is ties-to-even.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fp64_math.h"

namespace vk {

// ---------------- layouts ----------------
struct Layout {
    uint32_t n_draw_objects, n_paths, n_clips, bin_data_start;
    uint32_t path_tag_base, path_data_base, draw_tag_base, draw_data_base, transform_base, style_base;
};
struct Config {  // vello_encoding/src/config.rs:124-154
    uint32_t width_in_tiles, height_in_tiles, target_width, target_height, base_color;
    Layout layout;
    uint32_t lines_size, binning_size, tiles_size, seg_counts_size, segments_size, blend_size, ptcl_size;
};
struct Bump {  // config.rs:24-37
    uint32_t failed, binning, ptcl, tile, seg_counts, segments, blend, lines;
};
struct TagMonoid { uint32_t trans_ix, pathseg_ix, pathseg_offset, style_ix, path_ix; };
struct PathBbox { int32_t x0, y0, x1, y1; uint32_t draw_flags, trans_ix; };
struct LineSoup { uint32_t path_ix, pad; float p0x, p0y, p1x, p1y; };
struct SegmentCount { uint32_t line_ix, counts; };
struct Segment { float p0x, p0y, p1x, p1y, y_edge; uint32_t pad; };
struct Path { uint32_t bbox[4]; uint32_t tiles; uint32_t pad[3]; };
struct Tile { int32_t backdrop; uint32_t segment_count_or_ix; };
struct DrawMonoid { uint32_t path_ix, clip_ix, scene_offset, info_offset; };
struct Clip { uint32_t ix; int32_t path_ix; };
struct BinHeader { uint32_t element_count, chunk_offset; };
struct Bbox4 { float x0, y0, x1, y1; };
// engine-internal (not a reference layout): what coarse needs of one draw object, gathered once by k_coarse_prep
struct __attribute__((aligned(16))) CoarseEl { uint32_t tag, flags, w0, dd, di, tiles, bbox_x, bbox_y; };

static_assert(sizeof(Config) == 88, "ConfigUniform");
static_assert(sizeof(Bump) == 32, "BumpAllocators");
static_assert(sizeof(TagMonoid) == 20, "PathMonoid");
static_assert(sizeof(PathBbox) == 24, "PathBbox");
static_assert(sizeof(LineSoup) == 24, "LineSoup");
static_assert(sizeof(Segment) == 24, "PathSegment");
static_assert(sizeof(Path) == 32, "Path");
static_assert(sizeof(Tile) == 8, "Tile");
static_assert(sizeof(DrawMonoid) == 16, "DrawMonoid");
static_assert(sizeof(CoarseEl) == 32, "CoarseEl");

// ---------------- constants ----------------
constexpr uint32_t TILE_WIDTH = 16, TILE_HEIGHT = 16, N_TILE_X = 16, N_TILE_Y = 16, N_TILE = 256;
constexpr uint32_t PATH_TAG_SEG_TYPE = 3, PATH_TAG_LINETO = 1, PATH_TAG_QUADTO = 2, PATH_TAG_CUBICTO = 3;
constexpr uint32_t PATH_TAG_F32 = 8, PATH_TAG_TRANSFORM = 0x20, PATH_TAG_PATH = 0x10, PATH_TAG_STYLE = 0x40;
constexpr uint32_t PATH_TAG_SUBPATH_END = 4, STYLE_SIZE_IN_WORDS = 2;
constexpr uint32_t STYLE_FLAGS_STYLE = 0x80000000u, STYLE_FLAGS_FILL = 0x40000000u, STYLE_MITER_LIMIT_MASK = 0xFFFFu;
constexpr uint32_t STYLE_FLAGS_START_CAP_MASK = 0x0C000000u, STYLE_FLAGS_END_CAP_MASK = 0x03000000u;
constexpr uint32_t STYLE_FLAGS_CAP_SQUARE = 0x01000000u, STYLE_FLAGS_CAP_ROUND = 0x02000000u;
constexpr uint32_t STYLE_FLAGS_JOIN_MASK = 0x30000000u, STYLE_FLAGS_JOIN_BEVEL = 0u;
constexpr uint32_t STYLE_FLAGS_JOIN_MITER = 0x10000000u, STYLE_FLAGS_JOIN_ROUND = 0x20000000u;
constexpr uint32_t DRAWTAG_NOP = 0, DRAWTAG_FILL_COLOR = 0x44, DRAWTAG_FILL_LIN_GRADIENT = 0x114;
constexpr uint32_t DRAWTAG_FILL_RAD_GRADIENT = 0x29c, DRAWTAG_FILL_SWEEP_GRADIENT = 0x254, DRAWTAG_FILL_IMAGE = 0x28C;
constexpr uint32_t DRAWTAG_BLURRED_ROUNDED_RECT = 0x2d4, DRAWTAG_BEGIN_CLIP = 0x49, DRAWTAG_END_CLIP = 0x21;
constexpr uint32_t DRAW_INFO_FLAGS_FILL_RULE_BIT = 1;
constexpr uint32_t STAGE_BINNING = 0x1, STAGE_TILE_ALLOC = 0x2, STAGE_FLATTEN = 0x4, STAGE_PATH_COUNT = 0x8, STAGE_COARSE = 0x10;
constexpr uint32_t PTCL_INITIAL_ALLOC = 64, PTCL_INCREMENT = 256, PTCL_HEADROOM = 2;
constexpr uint32_t CMD_END = 0, CMD_FILL = 1, CMD_SOLID = 3, CMD_COLOR = 5, CMD_LIN_GRAD = 6, CMD_RAD_GRAD = 7;
constexpr uint32_t CMD_SWEEP_GRAD = 8, CMD_IMAGE = 9, CMD_BEGIN_CLIP = 10, CMD_END_CLIP = 11, CMD_JUMP = 12, CMD_BLUR_RECT = 13;
constexpr uint32_t BLEND_STACK_SPLIT = 4;
constexpr uint32_t RAD_GRAD_KIND_CIRCULAR = 1, RAD_GRAD_KIND_STRIP = 2, RAD_GRAD_KIND_FOCAL_ON_CIRCLE = 3, RAD_GRAD_KIND_CONE = 4;
constexpr uint32_t RAD_GRAD_SWAPPED = 1;
constexpr float ONE_MINUS_ULP = 0.99999994f;
constexpr float ROBUST_EPSILON = 2e-7f;

// ---------------- scalar helpers ----------------
// sin / cos: fp64_math.h (one branch-free path for every argument the flattener produces; ocml beyond it).
static __device__ __attribute__((noinline)) void sincos_large(float x, float &s, float &c) {  // cold: one copy of ocml's Payne-Hanek path
    double sd, cd;
    sincos((double)x, &sd, &cd);
    s = (float)sd;
    c = (float)cd;
}
__device__ __forceinline__ void sincos_cr(float x, float &s, float &c) {
    if (fabsf(x) <= (float)f64::SINCOS_MAX_ARG) {
        double sd, cd;
        f64::sincos_medium((double)x, sd, cd);
        s = (float)sd;
        c = (float)cd;
    } else {
        sincos_large(x, s, c);
    }
}
__device__ __forceinline__ float sin_cr(float x) {
    float s, c;
    sincos_cr(x, s, c);
    return s;
}
__device__ __forceinline__ float cos_cr(float x) {
    float s, c;
    sincos_cr(x, s, c);
    return c;
}
// Small-argument fast path: for |y / x| <= 2^-6 the truncated Taylor series evaluated in fp64 is accurate to
// < 2^-57 relative, i.e. it rounds to the same f32 as the full ocml routine (both are "the exact value
// rounded once" up to ~2^-29 per call) at a fifth of the instructions.
__device__ __forceinline__ float atan2_cr(float y, float x) {
    if (x > 0.0f && fabsf(y) <= 0.015625f * x && fabsf(y) < __builtin_inff()) {  // (inf, inf) is pi/4, not inf / inf
        double r = (double)y / (double)x;
        double r2 = r * r;
        return (float)(r * (1.0 + r2 * (-1.0 / 3.0 + r2 * (1.0 / 5.0 + r2 * (-1.0 / 7.0 + r2 * (1.0 / 9.0 + r2 * (-1.0 / 11.0)))))));
    }
    return (float)atan2((double)y, (double)x);
}
__device__ __forceinline__ float asin_cr(float x) { return (float)asin((double)x); }
__device__ __forceinline__ float acos_cr(float x) { return (float)acos((double)x); }
// (own fp64 kernel for the arguments the flattener produces -- a positive finite base, |y| <= 8: 115 instructions against
// ocml's 253, same rounded-once contract, fp64_math.h; everything else goes to ocml)
__device__ __forceinline__ float pow_cr(float x, float y) {
    if (x > 0.0f && x < __builtin_inff() && fabsf(y) <= 8.0f) return (float)f64::pow_pos((double)x, (double)y);
    return (float)pow((double)x, (double)y);
}
__device__ __forceinline__ float exp_cr(float x) { return (float)exp((double)x); }

// WGSL's u32(f32) / i32(f32): truncation, saturating, NaN -> 0.  That is exactly what v_cvt_u32_f32 / v_cvt_i32_f32 do in
// hardware (out-of-range values and infinities saturate, negative values give 0 unsigned, NaN gives 0), but a C cast of an
// out-of-range value is undefined, so the portable form has to guard it -- which on the GPU was three compares, three
// nested exec-mask branches and two constant moves around every conversion (k_fine's crossing-record loop: five
// conversions, a third of its instructions).  The instruction goes in as inline asm (not volatile: it is a pure function of
// its operand); scripts/calib/cvt_sat.hip checks it against the guarded form on the device for the values that matter.
__device__ __forceinline__ uint32_t f2u(float f) {
#ifdef VELLO_SIMT_EMU
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
#else
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#endif
}
__device__ __forceinline__ int32_t f2i(float f) {
#ifdef VELLO_SIMT_EMU
    if (f != f) return 0;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    if (f >= 2147483648.0f) return 0x7fffffff;
    return (int32_t)f;
#else
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#endif
}
__device__ __forceinline__ float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float minf(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float maxf(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }
__device__ __forceinline__ float roundf_te(float x) { return rintf(x); }
__device__ __forceinline__ int32_t mini(int32_t a, int32_t b) { return a < b ? a : b; }
__device__ __forceinline__ int32_t maxi(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t clampi(int32_t x, int32_t lo, int32_t hi) { return mini(maxi(x, lo), hi); }
__device__ __forceinline__ uint32_t minu(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t maxu(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t span(float a, float b) {
    return f2u(maxf(ceilf(maxf(a, b)) - floorf(minf(a, b)), 1.0f));
}

struct vec2 {
    float x, y;
};
__device__ __forceinline__ vec2 v2(float x, float y) { return vec2{x, y}; }
__device__ __forceinline__ vec2 operator+(vec2 a, vec2 b) { return vec2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ vec2 operator-(vec2 a, vec2 b) { return vec2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ vec2 operator*(vec2 a, float s) { return vec2{a.x * s, a.y * s}; }
__device__ __forceinline__ vec2 operator-(vec2 a) { return vec2{-a.x, -a.y}; }
__device__ __forceinline__ float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float length(vec2 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ vec2 normalize(vec2 a) {
    float l = length(a);
    return vec2{a.x / l, a.y / l};
}

struct Xform {
    float m0, m1, m2, m3, t0, t1;
};
// The index is u32 arithmetic as in WGSL (flatten.wgsl:306-316): a path encoded before any transform (the zero-width
// stroke clip of scene.rs:179-183 at the start of a scene) has trans_ix = 0 - 1 and reads the six words below
// transform_base, inside the scene buffer, instead of 24 GB past it.
__device__ __forceinline__ Xform read_transform(const uint32_t *scene, uint32_t base, uint32_t ix) {
    const uint32_t *p = scene + (uint32_t)(base + ix * 6u);
    return Xform{__uint_as_float(p[0]), __uint_as_float(p[1]), __uint_as_float(p[2]),
                 __uint_as_float(p[3]), __uint_as_float(p[4]), __uint_as_float(p[5])};
}

// ---------------- wave64 / workgroup primitives ----------------
constexpr int WAVE = 64;

// For workgroups of exactly ONE wave64 (k_fine): the LDS operations of a wave are issued and performed in program
// order, so lanes can exchange data through LDS without s_barrier and, above all, without the s_waitcnt that a
// __syncthreads() puts in front of it (it would also wait for the global loads in flight).  Only the compiler has
// to keep the order.  The CPU emulator runs lanes as fibers and needs the real rendezvous.
#ifdef VELLO_SIMT_EMU
// a rendezvous of the WAVE, not of the workgroup (waves of one workgroup may be on different tiles): the emulator's
// header provides it
#define wave_lds_sync() VELLO_EMU_WAVE_RENDEZVOUS()
#else
#define wave_lds_sync()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#endif

// v of the lane K to the left within a row of 16 lanes (lanes whose position in the row is < K get an unspecified value):
// a DPP row shift on the GPU -- a modifier of an ordinary VALU move instead of a trip through the LDS crossbar.
template <int K>
__device__ __forceinline__ uint32_t row_shr(uint32_t v) {
#ifdef VELLO_SIMT_EMU
    return (uint32_t)__shfl_up((int)v, K);
#else
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xf, 0xf, false);
#endif
}
// the same with 0 for the lanes whose source lies outside their row (DPP bound_ctrl): an operand modifier of the add that
// uses it, no register to preset
template <int K>
__device__ __forceinline__ uint32_t row_shr0(uint32_t v) {
#ifdef VELLO_SIMT_EMU
    const uint32_t o = (uint32_t)__shfl_up((int)v, K);
    return (threadIdx.x & 15u) >= (uint32_t)K ? o : 0u;
#else
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xf, 0xf, true);
#endif
}
// the top byte of v in all four bytes: v_perm_b32 (a shift and a 32-bit multiply otherwise)
__device__ __forceinline__ uint32_t bcast_byte3(uint32_t v) {
#ifdef VELLO_SIMT_EMU
    return (v >> 24) * 0x1010101u;
#else
    return __builtin_amdgcn_perm(v, v, 0x03030303u);
#endif
}
// v of a lane known at compile time (v_readlane on the GPU)
template <int L>
__device__ __forceinline__ uint32_t lane_value(uint32_t v) {
#ifdef VELLO_SIMT_EMU
    return (uint32_t)__shfl((int)v, L);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)v, L);
#endif
}

// A pointer known to point to global memory.  Through the parameter of an out-of-line function a pointer is generic, its
// loads are FLAT instructions, and a flat load counts on lgkmcnt as well as vmcnt: the counter that cannot be waited for
// out of order -- using the oldest of sixteen loads in flight waits for all sixteen.
#ifdef VELLO_SIMT_EMU
#define VK_GLOBAL
#else
#define VK_GLOBAL __attribute__((address_space(1)))
#endif
template <typename T>
__device__ __forceinline__ const VK_GLOBAL T *as_global(const T *p) {
    return (const VK_GLOBAL T *)p;
}

// The value, with its origin hidden from the optimiser: arithmetic on a loop-invariant (the lane number, say) is hoisted out
// of the loop and, when registers are short, SPILLED to scratch and reloaded -- dearer by far than the one or two VALU
// operations it saves.  What is derived from opaque(x) inside a loop stays there.
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
#ifndef VELLO_SIMT_EMU
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// v of lane `src` (0..63), any lane per lane: ds_bpermute with the byte address src * 4.  (hip's __shfl also folds its
// `width` argument in -- a lane-id v_mbcnt the compiler hoists out of every loop and then spills.)
__device__ __forceinline__ uint32_t wave_shfl(uint32_t v, uint32_t src) {
#ifdef VELLO_SIMT_EMU
    return (uint32_t)__shfl((int)v, (int)src);
#else
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
#endif
}
// v of the lane `src`, the same for every lane (a scalar): v_readlane
__device__ __forceinline__ uint32_t wave_read(uint32_t v, uint32_t src) {
#ifdef VELLO_SIMT_EMU
    return (uint32_t)__shfl((int)v, (int)src);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)src));
#endif
}

// number of set bits of a (wave-uniform) 64-bit mask below this lane: v_mbcnt_lo / _hi on the GPU
__device__ __forceinline__ uint32_t mask_rank_below(unsigned long long m, uint32_t lane) {
#ifdef VELLO_SIMT_EMU
    return (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
#else
    (void)lane;
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#endif
}

// Inclusive scans over the wave.  On the GPU: Hillis-Steele inside the rows of 16 lanes with DPP row shifts (a lane whose
// source lies outside its row gets 0), then the row totals by row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows
// 2 and 3 -- twelve VALU operations, no trip through the LDS crossbar (ds_bpermute per step: address, wait, select, add).
#ifndef VELLO_SIMT_EMU
#define VK_DPP0(v, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rows), 0xf, false))
#endif
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#ifdef VELLO_SIMT_EMU
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
#else
    (void)lane;
    v += VK_DPP0(v, 0x111, 0xf);
    v += VK_DPP0(v, 0x112, 0xf);
    v += VK_DPP0(v, 0x114, 0xf);
    v += VK_DPP0(v, 0x118, 0xf);
    v += VK_DPP0(v, 0x142, 0xa);
    v += VK_DPP0(v, 0x143, 0xc);
    return v;
#endif
}
// the same with max (values >= 0: the filler 0 is neutral)
__device__ __forceinline__ uint32_t wave_incl_scan_max_u32(uint32_t v, int lane) {
#ifdef VELLO_SIMT_EMU
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v = v > o ? v : o;
    }
    return v;
#else
    (void)lane;
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, VK_DPP0(v, 0x111, 0xf));
    v = mx(v, VK_DPP0(v, 0x112, 0xf));
    v = mx(v, VK_DPP0(v, 0x114, 0xf));
    v = mx(v, VK_DPP0(v, 0x118, 0xf));
    v = mx(v, VK_DPP0(v, 0x142, 0xa));
    v = mx(v, VK_DPP0(v, 0x143, 0xc));
    return v;
#endif
}

// Inclusive scan over a 256-thread workgroup (4 waves): wave shuffles + one LDS hop.
// `sh` needs 4 entries.  Returns the inclusive prefix; *total receives the workgroup sum.
__device__ __forceinline__ uint32_t block256_incl_scan_u32(uint32_t v, uint32_t *sh, uint32_t *total) {
    int tid = threadIdx.x;
    int lane = tid & 63, w = tid >> 6;
    uint32_t s = wave_incl_scan_u32(v, lane);
    __syncthreads();  // protect sh from a previous use
    if (lane == 63) sh[w] = s;
    __syncthreads();
    uint32_t s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3];
    uint32_t add = (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u);
    *total = s0 + s1 + s2 + s3;
    return s + add;
}

}  // namespace vk
