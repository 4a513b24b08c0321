/*
 * vello_oracle.h -- CPU restatement of the vello GPU compute pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vello_amd/ may include, link or
 * dlopen this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Every stage restates the reference algorithm on the reference's own byte
 * layouts (vello_encoding/src/{config,path,draw,clip,binning}.rs), following
 * vello_shaders/src/cpu/<stage>.rs cross-checked with
 * vello_shaders/shader/<stage>.wgsl; where the two disagree the WGSL wins
 * (SURVEY.md appendix D).  The fine stage has no CPU twin upstream and is
 * restated from vello_shaders/shader/fine.wgsl.
 *
 * Pinned against: vello_tests/snapshots/smoke/filled_{circle,square}.png,
 * vello_tests/tests/property.rs exact counts, mask.rs LUT hashes
 * (see tests/test_oracle_golden.py).
 */
#ifndef VELLO_ORACLE_H
#define VELLO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vello_encoding/src/resolve.rs:18-39 */
typedef struct {
    uint32_t n_draw_objects, n_paths, n_clips, bin_data_start;
    uint32_t path_tag_base, path_data_base, draw_tag_base, draw_data_base;
    uint32_t transform_base, style_base;
} vo_layout;

/* vello_encoding/src/config.rs:124-154 */
typedef struct {
    uint32_t width_in_tiles, height_in_tiles, target_width, target_height;
    uint32_t base_color;
    vo_layout layout;
    uint32_t lines_size, binning_size, tiles_size, seg_counts_size;
    uint32_t segments_size, blend_size, ptcl_size;
} vo_config;

/* vello_encoding/src/config.rs:24-37 */
typedef struct {
    uint32_t failed, binning, ptcl, tile, seg_counts, segments, blend, lines;
} vo_bump;

/* buffer ids for vo_buffer() */
enum {
    VO_BUF_TAG_MONOIDS = 0, /* 20 B each */
    VO_BUF_PATH_BBOXES,     /* 24 B */
    VO_BUF_BUMP,            /* 32 B */
    VO_BUF_LINES,           /* 24 B */
    VO_BUF_DRAW_MONOIDS,    /* 16 B */
    VO_BUF_INFO_BIN_DATA,   /* u32 */
    VO_BUF_CLIP_INP,        /* 8 B */
    VO_BUF_CLIP_BBOXES,     /* 16 B */
    VO_BUF_DRAW_BBOXES,     /* 16 B */
    VO_BUF_BIN_HEADERS,     /* 8 B */
    VO_BUF_PATHS,           /* 32 B */
    VO_BUF_TILES,           /* 8 B */
    VO_BUF_SEG_COUNTS,      /* 8 B */
    VO_BUF_SEGMENTS,        /* 24 B */
    VO_BUF_PTCL,            /* u32 */
    VO_BUF_BLEND_SPILL,     /* u32 */
    VO_BUF_OUTPUT,          /* RGBA8, w*h*4 */
    VO_BUF_COUNT
};

/* stage ids; vo_run executes [first, last] inclusive */
enum {
    VO_STAGE_PATHTAG_SCAN = 0, /* pathtag_reduce + pathtag_scan + bbox_clear */
    VO_STAGE_FLATTEN,
    VO_STAGE_DRAW_SCAN,        /* draw_reduce + draw_leaf */
    VO_STAGE_CLIP,             /* clip_reduce + clip_leaf */
    VO_STAGE_BINNING,
    VO_STAGE_TILE_ALLOC,
    VO_STAGE_PATH_COUNT,       /* path_count_setup + path_count */
    VO_STAGE_BACKDROP,
    VO_STAGE_COARSE,
    VO_STAGE_PATH_TILING,      /* path_tiling_setup + path_tiling */
    VO_STAGE_FINE,
    VO_STAGE_COUNT
};

enum { VO_AA_AREA = 0, VO_AA_MSAA8 = 1, VO_AA_MSAA16 = 2 };

typedef struct vo_ctx vo_ctx;

/* scale multiplies the reference's fixed bump capacities (config.rs:401-408);
 * 1 = reference sizes. */
vo_ctx *vo_create(uint32_t capacity_scale);
void vo_destroy(vo_ctx *);

/* Binds a packed scene (resolve.rs:107-154 layout) and computes the
 * ConfigUniform / buffer sizes exactly as RenderConfig::new
 * (config.rs:168-196).  The scene bytes are copied. */
int vo_set_scene(vo_ctx *, const uint8_t *scene, size_t scene_len,
                 const vo_layout *layout, uint32_t width, uint32_t height,
                 uint32_t base_color_premul_rgba8, int aa);

/* Optional gradient ramp texture (ramp_cache.rs: 512 RGBA8 texels per ramp). */
int vo_set_ramps(vo_ctx *, const uint32_t *ramps, uint32_t n_ramps);

/* Optional image atlas (render.rs:160-203: one Rgba8Unorm texture, images written at the xy the Resolver
 * patched into DrawImage).  `rgba8` holds width*height texels, row-major, 4 bytes each; copied. */
int vo_set_image_atlas(vo_ctx *, const uint8_t *rgba8, uint32_t width, uint32_t height);

const vo_config *vo_get_config(const vo_ctx *);

/* Runs stages first..last (inclusive).  Stage VO_STAGE_PATHTAG_SCAN also
 * zeroes the bump allocators (render.rs:313). */
int vo_run(vo_ctx *, int first_stage, int last_stage);

/* Full frame: vo_run(0, FINE); copies RGBA8 into out (w*h*4) if non-null. */
int vo_render(vo_ctx *, uint8_t *out_rgba8);

/* Raw access to intermediates (for differential tests).  Returns pointer and
 * writes the allocated size in bytes. */
void *vo_buffer(vo_ctx *, int buf_id, size_t *size_bytes);

/* mask.rs:36-98 */
void vo_make_mask_lut(uint8_t out[1024]);
void vo_make_mask_lut_16(uint8_t out[8192]);

/* CPU baseline leg: n_threads > 1 runs flatten (per tag), path_count (per line), coarse (per bin), path_tiling (per
 * crossing) and fine (per tile) on that many threads -- the units the shaders themselves run in parallel, with the
 * shaders' own atomics; outputs then differ from the serial run by what the order of atomics decides.  0/1 = serial,
 * the mode every parity test uses. */
void vo_set_threads(vo_ctx *, int n_threads);
/* pools = capacity_scale x config.rs:398-408's sizes from the next vo_set_scene on (growable pools: oracle.py auto_grow) */
void vo_set_capacity_scale(vo_ctx *c, uint32_t capacity_scale);
uint32_t vo_get_capacity_scale(const vo_ctx *c);

#ifdef __cplusplus
}
#endif
#endif
