/* Internal definitions for the CPU oracle (test infrastructure only). */
#ifndef VO_INTERNAL_H
#define VO_INTERNAL_H

#include "vello_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- POD layouts (vello_encoding/src/path.rs etc.) ---- */
typedef struct { uint32_t trans_ix, pathseg_ix, pathseg_offset, style_ix, path_ix; } vo_tag_monoid;      /* path.rs:321-332 */
typedef struct { int32_t x0, y0, x1, y1; uint32_t draw_flags, trans_ix; } vo_path_bbox;                  /* path.rs:382-395 */
typedef struct { uint32_t path_ix, pad; float p0[2], p1[2]; } vo_line_soup;                              /* path.rs:194-199 */
typedef struct { uint32_t line_ix, counts; } vo_seg_count;                                               /* path.rs:204-211 */
typedef struct { float p0[2], p1[2]; float y_edge; uint32_t pad; } vo_segment;                           /* path.rs:216-222 */
typedef struct { uint32_t bbox[4]; uint32_t tiles; uint32_t pad[3]; } vo_path;                           /* path.rs:404-410 */
typedef struct { int32_t backdrop; uint32_t segment_count_or_ix; } vo_tile;                              /* path.rs:415-423 */
typedef struct { uint32_t path_ix, clip_ix, scene_offset, info_offset; } vo_draw_monoid;                 /* draw.rs:241-250 */
typedef struct { uint32_t ix; int32_t path_ix; } vo_clip;                                                /* clip.rs:41-48 */
typedef struct { uint32_t element_count, chunk_offset; } vo_bin_header;                                  /* binning.rs:9-12 */

_Static_assert(sizeof(vo_config) == 88, "ConfigUniform");
_Static_assert(sizeof(vo_bump) == 32, "BumpAllocators");
_Static_assert(sizeof(vo_tag_monoid) == 20, "PathMonoid");
_Static_assert(sizeof(vo_path_bbox) == 24, "PathBbox");
_Static_assert(sizeof(vo_line_soup) == 24, "LineSoup");
_Static_assert(sizeof(vo_segment) == 24, "PathSegment");
_Static_assert(sizeof(vo_path) == 32, "Path");
_Static_assert(sizeof(vo_tile) == 8, "Tile");

/* ---- constants ---- */
#define VO_WG 256u
#define TILE_WIDTH 16u
#define TILE_HEIGHT 16u
#define N_TILE_X 16u
#define N_TILE_Y 16u
#define N_TILE 256u

/* shared/pathtag.wgsl */
#define PATH_TAG_SEG_TYPE 3u
#define PATH_TAG_LINETO 1u
#define PATH_TAG_QUADTO 2u
#define PATH_TAG_CUBICTO 3u
#define PATH_TAG_F32 8u
#define PATH_TAG_TRANSFORM 0x20u
#define PATH_TAG_PATH 0x10u
#define PATH_TAG_STYLE 0x40u
#define PATH_TAG_SUBPATH_END 4u
#define STYLE_SIZE_IN_WORDS 2u
#define STYLE_FLAGS_STYLE 0x80000000u
#define STYLE_FLAGS_FILL 0x40000000u
#define STYLE_MITER_LIMIT_MASK 0xFFFFu
#define STYLE_FLAGS_START_CAP_MASK 0x0C000000u
#define STYLE_FLAGS_END_CAP_MASK 0x03000000u
#define STYLE_FLAGS_CAP_BUTT 0u
#define STYLE_FLAGS_CAP_SQUARE 0x01000000u
#define STYLE_FLAGS_CAP_ROUND 0x02000000u
#define STYLE_FLAGS_JOIN_MASK 0x30000000u
#define STYLE_FLAGS_JOIN_BEVEL 0u
#define STYLE_FLAGS_JOIN_MITER 0x10000000u
#define STYLE_FLAGS_JOIN_ROUND 0x20000000u

/* shared/drawtag.wgsl */
#define DRAWTAG_NOP 0u
#define DRAWTAG_FILL_COLOR 0x44u
#define DRAWTAG_FILL_LIN_GRADIENT 0x114u
#define DRAWTAG_FILL_RAD_GRADIENT 0x29cu
#define DRAWTAG_FILL_SWEEP_GRADIENT 0x254u
#define DRAWTAG_FILL_IMAGE 0x28Cu
#define DRAWTAG_BLURRED_ROUNDED_RECT 0x2d4u
#define DRAWTAG_BEGIN_CLIP 0x49u
#define DRAWTAG_END_CLIP 0x21u
#define DRAW_INFO_FLAGS_FILL_RULE_BIT 1u

/* shared/bump.wgsl */
#define STAGE_BINNING 0x1u
#define STAGE_TILE_ALLOC 0x2u
#define STAGE_FLATTEN 0x4u
#define STAGE_PATH_COUNT 0x8u
#define STAGE_COARSE 0x10u

/* shared/ptcl.wgsl */
#define PTCL_INITIAL_ALLOC 64u
#define PTCL_INCREMENT 256u
#define PTCL_HEADROOM 2u
#define CMD_END 0u
#define CMD_FILL 1u
#define CMD_SOLID 3u
#define CMD_COLOR 5u
#define CMD_LIN_GRAD 6u
#define CMD_RAD_GRAD 7u
#define CMD_SWEEP_GRAD 8u
#define CMD_IMAGE 9u
#define CMD_BEGIN_CLIP 10u
#define CMD_END_CLIP 11u
#define CMD_JUMP 12u
#define CMD_BLUR_RECT 13u
#define BLEND_STACK_SPLIT 4u

#define RAD_GRAD_KIND_CIRCULAR 1u
#define RAD_GRAD_KIND_STRIP 2u
#define RAD_GRAD_KIND_FOCAL_ON_CIRCLE 3u
#define RAD_GRAD_KIND_CONE 4u
#define RAD_GRAD_SWAPPED 1u

/* cpu/util.rs:215-228 */
#define ONE_MINUS_ULP 0.99999994f
#define ROBUST_EPSILON 2e-7f

struct vo_ctx {
    uint32_t cap_scale;
    uint32_t *scene;
    size_t scene_words;
    vo_config cfg;
    int aa;
    int n_threads;
/* threads of the OpenMP stages: their shared bump counters and tile words are plain atomics, which stop scaling long
 * before a 256-thread host is used up (measured: slower at 256 than at 8); fine, with no shared writes, takes them all */
#define VO_OMP_THREADS(c) ((c)->n_threads > 32 ? 32 : ((c)->n_threads > 1 ? (c)->n_threads : 1))
    uint32_t n_tag_words; /* padded tag bytes / 4 */
    uint32_t n_ramps;
    uint32_t *ramps;
    uint32_t atlas_w, atlas_h;
    uint32_t *atlas;
    void *buf[VO_BUF_COUNT];
    size_t buf_size[VO_BUF_COUNT];
    uint8_t mask_lut8[1024];
    uint8_t mask_lut16[8192];
};

/* ---- numeric helpers shared by all stages ----
 *
 * f32 transcendentals are defined as the f64 libm value rounded to f32.  WGSL
 * leaves their precision implementation-defined and Rust's f32 methods call
 * the platform libm; "round-to-nearest of the exact value" is the ideal both
 * approximate, and it is reproducible on the GPU (fp64 ocml + one rounding).
 */
static inline float vo_sinf(float x) { return (float)sin((double)x); }
static inline float vo_cosf(float x) { return (float)cos((double)x); }
static inline float vo_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
static inline float vo_asinf(float x) { return (float)asin((double)x); }
static inline float vo_acosf(float x) { return (float)acos((double)x); }
static inline float vo_powf(float x, float y) { return (float)pow((double)x, (double)y); }
static inline float vo_expf(float x) { return (float)exp((double)x); }

static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* WGSL u32(f32)/i32(f32) are saturating; plain C casts are UB out of range. */
static inline uint32_t f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
static inline int32_t f2i(float f) {
    if (f != f) return 0;
    if (f <= -2147483648.0f) return INT32_MIN;
    if (f >= 2147483648.0f) return INT32_MAX;
    return (int32_t)f;
}
static inline float vo_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline float vo_min(float a, float b) { return a < b ? a : b; }
static inline float vo_max(float a, float b) { return a > b ? a : b; }
static inline float vo_clamp(float x, float lo, float hi) { return vo_min(vo_max(x, lo), hi); }
/* WGSL round(): ties to even */
static inline float vo_round(float x) { return rintf(x); }
static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static inline int32_t iclamp(int32_t x, int32_t lo, int32_t hi) { return imin(imax(x, lo), hi); }
static inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline uint32_t popcnt(uint32_t x) { return (uint32_t)__builtin_popcount(x); }

/* path_count.wgsl:38-40 */
static inline uint32_t vo_span(float a, float b) {
    return f2u(vo_max(ceilf(vo_max(a, b)) - floorf(vo_min(a, b)), 1.0f));
}

typedef struct { float x, y; } vec2;
static inline vec2 v2(float x, float y) { vec2 r = {x, y}; return r; }
static inline vec2 vadd(vec2 a, vec2 b) { return v2(a.x + b.x, a.y + b.y); }
static inline vec2 vsub(vec2 a, vec2 b) { return v2(a.x - b.x, a.y - b.y); }
static inline vec2 vmul(vec2 a, float s) { return v2(a.x * s, a.y * s); }
static inline vec2 vneg(vec2 a) { return v2(-a.x, -a.y); }
static inline float vdot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
/* WGSL length() = sqrt(dot(v,v)) */
static inline float vlen(vec2 a) { return sqrtf(vdot(a, a)); }
static inline vec2 vnorm(vec2 a) { float l = vlen(a); return v2(a.x / l, a.y / l); }

typedef struct { float m[4]; float t[2]; } vo_xform;
static inline vo_xform vo_read_transform(const uint32_t *scene, uint32_t base, uint32_t ix) {
    vo_xform r;
    const uint32_t *p = scene + (uint32_t)(base + ix * 6u); /* u32 index arithmetic, as WGSL: trans_ix may be 0 - 1 */
    for (int i = 0; i < 4; i++) r.m[i] = bits2f(p[i]);
    r.t[0] = bits2f(p[4]);
    r.t[1] = bits2f(p[5]);
    return r;
}

/* stage entry points */
void vo_stage_pathtag_scan(vo_ctx *c);
void vo_stage_flatten(vo_ctx *c);
void vo_stage_draw_scan(vo_ctx *c);
void vo_stage_clip(vo_ctx *c);
void vo_stage_binning(vo_ctx *c);
void vo_stage_tile_alloc(vo_ctx *c);
void vo_stage_path_count(vo_ctx *c);
void vo_stage_backdrop(vo_ctx *c);
void vo_stage_coarse(vo_ctx *c);
void vo_stage_path_tiling(vo_ctx *c);
void vo_stage_fine(vo_ctx *c);

vo_tag_monoid vo_reduce_tag(uint32_t tag_word);

#endif
