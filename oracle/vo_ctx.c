/*
 * Oracle context: scene binding, RenderConfig (config.rs:168-196, :363-435),
 * stage driver (render.rs:250-502, :560-616 order), mask LUTs (mask.rs).
 * TEST INFRASTRUCTURE ONLY (see vello_oracle.h).
 */
#include "vo_internal.h"

static uint32_t align_up(uint32_t len, uint32_t alignment) { return len + ((0u - len) & (alignment - 1u)); }

vo_ctx *vo_create(uint32_t capacity_scale) {
    vo_ctx *c = (vo_ctx *)calloc(1, sizeof(vo_ctx));
    if (!c) return NULL;
    c->cap_scale = capacity_scale ? capacity_scale : 1u;
    c->n_threads = 1;
    vo_make_mask_lut(c->mask_lut8);
    vo_make_mask_lut_16(c->mask_lut16);
    return c;
}

void vo_destroy(vo_ctx *c) {
    if (!c) return;
    for (int i = 0; i < VO_BUF_COUNT; i++) free(c->buf[i]);
    free(c->scene);
    free(c->ramps);
    free(c->atlas);
    free(c);
}

void vo_set_threads(vo_ctx *c, int n) { c->n_threads = n; }

/* Pool sizes = capacity_scale x the reference's fixed ones (config.rs:398-408); takes effect at the next vo_set_scene.
 * The Python wrapper's auto_grow doubles it and renders again while a frame overflows a pool. */
void vo_set_capacity_scale(vo_ctx *c, uint32_t capacity_scale) { c->cap_scale = capacity_scale ? capacity_scale : 1u; }
uint32_t vo_get_capacity_scale(const vo_ctx *c) { return c->cap_scale; }

static int ensure(vo_ctx *c, int id, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (c->buf_size[id] >= bytes) return 0;
    free(c->buf[id]);
    c->buf[id] = calloc(1, bytes + 64);
    c->buf_size[id] = c->buf[id] ? bytes : 0;
    return c->buf[id] ? 0 : -1;
}

int vo_set_scene(vo_ctx *c, const uint8_t *scene, size_t scene_len, const vo_layout *layout, uint32_t width,
                 uint32_t height, uint32_t base_color, int aa) {
    if (!c || (!scene && scene_len) || !layout || (scene_len & 3u)) return -1;
    free(c->scene);
    /* slack so that speculative reads just past the tag stream stay in bounds */
    c->scene = (uint32_t *)calloc(1, scene_len + 64);
    if (!c->scene) return -1;
    if (scene_len) memcpy(c->scene, scene, scene_len);
    c->scene_words = scene_len / 4u;
    c->aa = aa;
    vo_config *g = &c->cfg;
    memset(g, 0, sizeof *g);
    /* RenderConfig::new, config.rs:168-196 */
    uint32_t new_width = align_up(width, TILE_WIDTH);
    uint32_t new_height = align_up(height, TILE_HEIGHT);
    g->width_in_tiles = new_width / TILE_WIDTH;
    g->height_in_tiles = new_height / TILE_HEIGHT;
    g->target_width = width;
    g->target_height = height;
    g->base_color = base_color;
    g->layout = *layout;
    /* BufferSizes::new, config.rs:363-435 */
    uint32_t n_paths = layout->n_paths, n_draw = layout->n_draw_objects, n_clips = layout->n_clips;
    uint32_t n_path_tags = (layout->path_data_base - layout->path_tag_base) * 4u;
    uint32_t path_tag_padded = align_up(n_path_tags, 4u * VO_WG);
    c->n_tag_words = path_tag_padded / 4u;
    uint32_t binning_wgs = (n_draw + VO_WG - 1u) / VO_WG;
    uint32_t width_in_bins = (g->width_in_tiles + 15u) / 16u, height_in_bins = (g->height_in_tiles + 15u) / 16u;
    uint32_t aligned_n_bins = align_up(width_in_bins * height_in_bins, 256u);
    uint32_t s = c->cap_scale;
    uint32_t bin_data = (1u << 18) * s, tiles = (1u << 21) * s, lines = (1u << 21) * s;
    uint32_t seg_counts = (1u << 21) * s, segments = (1u << 21) * s, blend = (1u << 20) * s, ptcl = (1u << 23) * s;
    if (bin_data <= layout->bin_data_start) bin_data = layout->bin_data_start + 1u;
    g->lines_size = lines;
    g->binning_size = bin_data - layout->bin_data_start;
    g->tiles_size = tiles;
    g->seg_counts_size = seg_counts;
    g->segments_size = segments;
    g->blend_size = blend;
    g->ptcl_size = ptcl;
    int e = 0;
    e |= ensure(c, VO_BUF_TAG_MONOIDS, (size_t)(c->n_tag_words + 1u) * sizeof(vo_tag_monoid));
    e |= ensure(c, VO_BUF_PATH_BBOXES, (size_t)(n_paths + 1u) * sizeof(vo_path_bbox));
    e |= ensure(c, VO_BUF_BUMP, sizeof(vo_bump));
    e |= ensure(c, VO_BUF_LINES, (size_t)lines * sizeof(vo_line_soup));
    e |= ensure(c, VO_BUF_DRAW_MONOIDS, (size_t)(n_draw + 1u) * sizeof(vo_draw_monoid));
    e |= ensure(c, VO_BUF_INFO_BIN_DATA, (size_t)bin_data * 4u);
    e |= ensure(c, VO_BUF_CLIP_INP, (size_t)(n_clips + 1u) * sizeof(vo_clip));
    e |= ensure(c, VO_BUF_CLIP_BBOXES, (size_t)(n_clips + 1u) * 16u);
    e |= ensure(c, VO_BUF_DRAW_BBOXES, (size_t)(n_paths + 1u) * 16u);
    e |= ensure(c, VO_BUF_BIN_HEADERS, (size_t)(binning_wgs * aligned_n_bins + 1u) * sizeof(vo_bin_header));
    e |= ensure(c, VO_BUF_PATHS, (size_t)(align_up(n_paths, 256u) + 256u) * sizeof(vo_path));
    e |= ensure(c, VO_BUF_TILES, (size_t)tiles * sizeof(vo_tile));
    e |= ensure(c, VO_BUF_SEG_COUNTS, (size_t)seg_counts * sizeof(vo_seg_count));
    e |= ensure(c, VO_BUF_SEGMENTS, (size_t)segments * sizeof(vo_segment));
    e |= ensure(c, VO_BUF_PTCL, (size_t)ptcl * 4u);
    e |= ensure(c, VO_BUF_BLEND_SPILL, (size_t)blend * 4u);
    e |= ensure(c, VO_BUF_OUTPUT, (size_t)width * height * 4u);
    return e;
}

int vo_set_image_atlas(vo_ctx *c, const uint8_t *rgba8, uint32_t width, uint32_t height) {
    free(c->atlas);
    c->atlas = NULL;
    c->atlas_w = c->atlas_h = 0;
    if (!rgba8 || !width || !height) return 0;
    c->atlas = (uint32_t *)malloc((size_t)width * height * 4u);
    if (!c->atlas) return -1;
    memcpy(c->atlas, rgba8, (size_t)width * height * 4u);
    c->atlas_w = width;
    c->atlas_h = height;
    return 0;
}

int vo_set_ramps(vo_ctx *c, const uint32_t *ramps, uint32_t n_ramps) {
    free(c->ramps);
    c->ramps = NULL;
    c->n_ramps = 0;
    if (!ramps || !n_ramps) return 0;
    c->ramps = (uint32_t *)malloc((size_t)n_ramps * 512u * 4u);
    if (!c->ramps) return -1;
    memcpy(c->ramps, ramps, (size_t)n_ramps * 512u * 4u);
    c->n_ramps = n_ramps;
    return 0;
}

const vo_config *vo_get_config(const vo_ctx *c) { return &c->cfg; }

void *vo_buffer(vo_ctx *c, int id, size_t *size_bytes) {
    if (id < 0 || id >= VO_BUF_COUNT) return NULL;
    if (size_bytes) *size_bytes = c->buf_size[id];
    return c->buf[id];
}

int vo_run(vo_ctx *c, int first, int last) {
    if (!c || !c->scene || first < 0 || last >= VO_STAGE_COUNT) return -1;
    for (int s = first; s <= last; s++) {
        switch (s) {
        case VO_STAGE_PATHTAG_SCAN: vo_stage_pathtag_scan(c); break;
        case VO_STAGE_FLATTEN: vo_stage_flatten(c); break;
        case VO_STAGE_DRAW_SCAN: vo_stage_draw_scan(c); break;
        case VO_STAGE_CLIP: vo_stage_clip(c); break;
        case VO_STAGE_BINNING: vo_stage_binning(c); break;
        case VO_STAGE_TILE_ALLOC: vo_stage_tile_alloc(c); break;
        case VO_STAGE_PATH_COUNT: vo_stage_path_count(c); break;
        case VO_STAGE_BACKDROP: vo_stage_backdrop(c); break;
        case VO_STAGE_COARSE: vo_stage_coarse(c); break;
        case VO_STAGE_PATH_TILING: vo_stage_path_tiling(c); break;
        case VO_STAGE_FINE: vo_stage_fine(c); break;
        default: return -1;
        }
    }
    return 0;
}

int vo_render(vo_ctx *c, uint8_t *out) {
    int r = vo_run(c, 0, VO_STAGE_FINE);
    if (r) return r;
    if (out) memcpy(out, c->buf[VO_BUF_OUTPUT], (size_t)c->cfg.target_width * c->cfg.target_height * 4u);
    const vo_bump *b = (const vo_bump *)c->buf[VO_BUF_BUMP];
    return b->failed ? 1 : 0;
}

/* ---- mask LUTs: vello_encoding/src/mask.rs:11-98 (f64 arithmetic) ---- */
static const uint8_t PATTERN[8] = {0, 5, 3, 7, 1, 4, 6, 2};
static const uint8_t PATTERN_16[16] = {1, 8, 4, 11, 15, 7, 3, 12, 0, 9, 5, 13, 2, 10, 6, 14};

static uint32_t one_mask_n(double slope, double translation, int is_pos, const uint8_t *pat, int n) {
    if (is_pos) translation = 1. - translation;
    uint32_t result = 0;
    double inv = 1.0 / (double)n;
    for (int i = 0; i < n; i++) {
        double y = ((double)i + 0.5) * inv;
        double x = ((double)pat[i] + 0.5) * inv;
        if (!is_pos) y = 1. - y;
        if ((x - (1.0 - translation)) * (1. - slope) - (y - translation) * slope >= 0.) result |= 1u << i;
    }
    return result;
}

void vo_make_mask_lut(uint8_t out[1024]) {
    const int W = 32, H = 32, HALF = 16;
    for (int i = 0; i < W * H; i++) {
        int u = i % W, v = i / W;
        int is_pos = v >= HALF;
        double y = ((double)(v % HALF) + 0.5) * (1.0 / (double)HALF);
        double x = ((double)u + 0.5) * (1.0 / (double)W);
        out[i] = (uint8_t)one_mask_n(y, x, is_pos, PATTERN, 8);
    }
}

void vo_make_mask_lut_16(uint8_t out[8192]) {
    const int W = 64, H = 64, HALF = 32;
    for (int i = 0; i < W * H; i++) {
        int u = i % W, v = i / W;
        int is_pos = v >= HALF;
        double y = ((double)(v % HALF) + 0.5) * (1.0 / (double)HALF);
        double x = ((double)u + 0.5) * (1.0 / (double)W);
        uint32_t m = one_mask_n(y, x, is_pos, PATTERN_16, 16);
        out[2 * i] = (uint8_t)(m & 0xff);
        out[2 * i + 1] = (uint8_t)(m >> 8);
    }
}
