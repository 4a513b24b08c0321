/*
 * Oracle middle: binning, tile_alloc, path_count, backdrop, coarse, path_tiling.
 * TEST INFRASTRUCTURE ONLY (see vello_oracle.h).
 */
#include "vo_internal.h"

/* ------------------------------------------------------------------ */
/* binning: shader/binning.wgsl:55-203, cpu/binning.rs:27-107          */
/* ------------------------------------------------------------------ */
void vo_stage_binning(vo_ctx *c) {
    const vo_config *cfg = &c->cfg;
    vo_bump *bump = (vo_bump *)c->buf[VO_BUF_BUMP];
    if (bump->lines > cfg->lines_size) { /* binning.wgsl:64-75 */
        bump->failed |= STAGE_FLATTEN;
        return;
    }
    const vo_draw_monoid *draw_monoids = (const vo_draw_monoid *)c->buf[VO_BUF_DRAW_MONOIDS];
    const vo_path_bbox *path_bbox_buf = (const vo_path_bbox *)c->buf[VO_BUF_PATH_BBOXES];
    const float(*clip_bbox_buf)[4] = (const float(*)[4])c->buf[VO_BUF_CLIP_BBOXES];
    float(*intersected_bbox)[4] = (float(*)[4])c->buf[VO_BUF_DRAW_BBOXES];
    uint32_t *bin_data = (uint32_t *)c->buf[VO_BUF_INFO_BIN_DATA];
    vo_bin_header *bin_header = (vo_bin_header *)c->buf[VO_BUF_BIN_HEADERS];
    const float SX = 1.0f / (float)(N_TILE_X * TILE_WIDTH);
    const float SY = 1.0f / (float)(N_TILE_Y * TILE_HEIGHT);
    int32_t width_in_bins = (int32_t)((cfg->width_in_tiles + N_TILE_X - 1u) / N_TILE_X);
    int32_t height_in_bins = (int32_t)((cfg->height_in_tiles + N_TILE_Y - 1u) / N_TILE_Y);
    uint32_t n_bins = (uint32_t)(width_in_bins * height_in_bins);
    uint32_t aligned_n_bins = (n_bins + N_TILE - 1u) & ~(N_TILE - 1u);
    uint32_t n_draw = cfg->layout.n_draw_objects;
    uint32_t n_wg = (n_draw + VO_WG - 1u) / VO_WG;
    uint32_t *counts = (uint32_t *)calloc(aligned_n_bins ? aligned_n_bins : 1, sizeof(uint32_t));
    uint32_t *chunk_offset = (uint32_t *)calloc(aligned_n_bins ? aligned_n_bins : 1, sizeof(uint32_t));
    int32_t(*bboxes)[4] = (int32_t(*)[4])malloc(sizeof(int32_t[4]) * VO_WG);
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        memset(counts, 0, sizeof(uint32_t) * aligned_n_bins);
        for (uint32_t local_ix = 0; local_ix < VO_WG; local_ix++) {
            uint32_t element_ix = wg * VO_WG + local_ix;
            int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
            if (element_ix < n_draw) {
                vo_draw_monoid dm = draw_monoids[element_ix];
                float clip_bbox[4] = {-1e9f, -1e9f, 1e9f, 1e9f};
                if (dm.clip_ix > 0u) {
                    uint32_t ci = umin(dm.clip_ix - 1u, cfg->layout.n_clips - 1u);
                    memcpy(clip_bbox, clip_bbox_buf[ci], sizeof clip_bbox);
                }
                vo_path_bbox pb = path_bbox_buf[dm.path_ix];
                float pbf[4] = {(float)pb.x0, (float)pb.y0, (float)pb.x1, (float)pb.y1};
                float bbox[4] = {vo_max(clip_bbox[0], pbf[0]), vo_max(clip_bbox[1], pbf[1]),
                                 vo_min(clip_bbox[2], pbf[2]), vo_min(clip_bbox[3], pbf[3])};
                memcpy(intersected_bbox[element_ix], bbox, sizeof bbox);
                if (bbox[0] < bbox[2] && bbox[1] < bbox[3]) {
                    x0 = f2i(floorf(bbox[0] * SX));
                    y0 = f2i(floorf(bbox[1] * SY));
                    x1 = f2i(ceilf(bbox[2] * SX));
                    y1 = f2i(ceilf(bbox[3] * SY));
                }
            }
            x0 = iclamp(x0, 0, width_in_bins);
            y0 = iclamp(y0, 0, height_in_bins);
            x1 = iclamp(x1, 0, width_in_bins);
            y1 = iclamp(y1, 0, height_in_bins);
            if (x0 == x1) y1 = y0;
            for (int32_t y = y0; y < y1; y++)
                for (int32_t x = x0; x < x1; x++) counts[y * width_in_bins + x]++;
            bboxes[local_ix][0] = x0; bboxes[local_ix][1] = y0; bboxes[local_ix][2] = x1; bboxes[local_ix][3] = y1;
        }
        for (uint32_t b = 0; b < aligned_n_bins; b++) {
            uint32_t element_count = counts[b];
            uint32_t co = bump->binning;
            bump->binning += element_count;
            if (co + element_count > cfg->binning_size) {
                co = 0u;
                bump->failed |= STAGE_BINNING;
            }
            chunk_offset[b] = co;
            bin_header[wg * aligned_n_bins + b].element_count = element_count;
            bin_header[wg * aligned_n_bins + b].chunk_offset = co;
        }
        for (uint32_t local_ix = 0; local_ix < VO_WG; local_ix++) {
            uint32_t element_ix = wg * VO_WG + local_ix;
            const int32_t *bb = bboxes[local_ix];
            for (int32_t y = bb[1]; y < bb[3]; y++) {
                for (int32_t x = bb[0]; x < bb[2]; x++) {
                    uint32_t bin_ix = (uint32_t)(y * width_in_bins + x);
                    uint32_t ix = cfg->layout.bin_data_start + chunk_offset[bin_ix];
                    if (chunk_offset[bin_ix] < cfg->binning_size) bin_data[ix] = element_ix;
                    chunk_offset[bin_ix]++;
                }
            }
        }
    }
    free(counts);
    free(chunk_offset);
    free(bboxes);
}

/* ------------------------------------------------------------------ */
/* tile_alloc: shader/tile_alloc.wgsl:35-123, cpu/tile_alloc.rs:14-56  */
/* ------------------------------------------------------------------ */
void vo_stage_tile_alloc(vo_ctx *c) {
    const vo_config *cfg = &c->cfg;
    vo_bump *bump = (vo_bump *)c->buf[VO_BUF_BUMP];
    if ((bump->failed & (STAGE_BINNING | STAGE_FLATTEN)) != 0u) return;
    const uint32_t *scene = c->scene;
    const float(*draw_bboxes)[4] = (const float(*)[4])c->buf[VO_BUF_DRAW_BBOXES];
    vo_path *paths = (vo_path *)c->buf[VO_BUF_PATHS];
    vo_tile *tiles = (vo_tile *)c->buf[VO_BUF_TILES];
    const float SX = 1.0f / (float)TILE_WIDTH, SY = 1.0f / (float)TILE_HEIGHT;
    uint32_t n_draw = cfg->layout.n_draw_objects;
    uint32_t n_wg = (cfg->layout.n_paths + VO_WG - 1u) / VO_WG; /* config.rs:251,266 */
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        uint32_t prefix[VO_WG];
        uint32_t bb[VO_WG][4];
        uint32_t total = 0;
        for (uint32_t l = 0; l < VO_WG; l++) {
            uint32_t drawobj_ix = wg * VO_WG + l;
            uint32_t drawtag = DRAWTAG_NOP;
            if (drawobj_ix < n_draw) drawtag = scene[cfg->layout.draw_tag_base + drawobj_ix];
            int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
            if (drawtag != DRAWTAG_NOP && drawtag != DRAWTAG_END_CLIP) {
                const float *bbox = draw_bboxes[drawobj_ix];
                if (bbox[0] < bbox[2] && bbox[1] < bbox[3]) {
                    x0 = f2i(floorf(bbox[0] * SX));
                    y0 = f2i(floorf(bbox[1] * SY));
                    x1 = f2i(ceilf(bbox[2] * SX));
                    y1 = f2i(ceilf(bbox[3] * SY));
                }
            }
            uint32_t ux0 = (uint32_t)iclamp(x0, 0, (int32_t)cfg->width_in_tiles);
            uint32_t uy0 = (uint32_t)iclamp(y0, 0, (int32_t)cfg->height_in_tiles);
            uint32_t ux1 = (uint32_t)iclamp(x1, 0, (int32_t)cfg->width_in_tiles);
            uint32_t uy1 = (uint32_t)iclamp(y1, 0, (int32_t)cfg->height_in_tiles);
            bb[l][0] = ux0; bb[l][1] = uy0; bb[l][2] = ux1; bb[l][3] = uy1;
            prefix[l] = total; /* exclusive */
            total += (ux1 - ux0) * (uy1 - uy0);
        }
        uint32_t offset = bump->tile;
        bump->tile += total;
        if (offset + total > cfg->tiles_size) {
            offset = 0u;
            bump->failed |= STAGE_TILE_ALLOC;
        }
        for (uint32_t l = 0; l < VO_WG; l++) {
            uint32_t drawobj_ix = wg * VO_WG + l;
            if (drawobj_ix < n_draw) {
                vo_path p;
                memset(&p, 0, sizeof p);
                memcpy(p.bbox, bb[l], sizeof p.bbox);
                p.tiles = offset + prefix[l];
                paths[drawobj_ix] = p;
            }
        }
        for (uint32_t i = 0; i < total && offset + i < cfg->tiles_size; i++) {
            tiles[offset + i].backdrop = 0;
            tiles[offset + i].segment_count_or_ix = 0u;
        }
    }
}

/* ------------------------------------------------------------------ */
/* path_count_setup + path_count: shader/path_count_setup.wgsl:17-27,  */
/* shader/path_count.wgsl:51-202 (WGSL comparisons), cpu/path_count.rs */
/* ------------------------------------------------------------------ */
void vo_stage_path_count(vo_ctx *c) {
    const vo_config *cfg = &c->cfg;
    vo_bump *bump = (vo_bump *)c->buf[VO_BUF_BUMP];
    if (bump->failed != 0u) return; /* path_count_setup.wgsl:18-19 */
    const vo_line_soup *lines = (const vo_line_soup *)c->buf[VO_BUF_LINES];
    const vo_path *paths = (const vo_path *)c->buf[VO_BUF_PATHS];
    vo_tile *tile = (vo_tile *)c->buf[VO_BUF_TILES];
    vo_seg_count *seg_counts = (vo_seg_count *)c->buf[VO_BUF_SEG_COUNTS];
    const float TILE_SCALE = 0.0625f;
    uint32_t n_lines = bump->lines;
    /* one invocation per line; tile counters and the SegmentCount bump are atomics in the shader (path_count.wgsl:172-199) */
#pragma omp parallel for schedule(dynamic, 4096) if (c->n_threads > 1) num_threads(VO_OMP_THREADS(c))
    for (uint32_t line_ix = 0; line_ix < n_lines; line_ix++) {
        vo_line_soup line = lines[line_ix];
        int is_down = line.p1[1] >= line.p0[1];
        vec2 xy0 = is_down ? v2(line.p0[0], line.p0[1]) : v2(line.p1[0], line.p1[1]);
        vec2 xy1 = is_down ? v2(line.p1[0], line.p1[1]) : v2(line.p0[0], line.p0[1]);
        vec2 s0 = vmul(xy0, TILE_SCALE);
        vec2 s1 = vmul(xy1, TILE_SCALE);
        uint32_t count_x = vo_span(s0.x, s1.x) - 1u;
        uint32_t count = count_x + vo_span(s0.y, s1.y);
        float dx = fabsf(s1.x - s0.x);
        float dy = s1.y - s0.y;
        if (dx + dy == 0.0f) continue;
        if (dy == 0.0f && floorf(s0.y) == s0.y) continue;
        float idxdy = 1.0f / (dx + dy);
        float a = dx * idxdy;
        int is_positive_slope = s1.x >= s0.x;
        float x_sign = is_positive_slope ? 1.0f : -1.0f;
        float xt0 = floorf(s0.x * x_sign);
        float cc = s0.x * x_sign - xt0;
        float y0 = floorf(s0.y);
        float ytop = (s0.y == s1.y) ? ceilf(s0.y) : y0 + 1.0f;
        float b = vo_min((dy * cc + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
        float robust_err = floorf(a * ((float)count - 1.0f) + b) - (float)count_x;
        if (robust_err != 0.0f) a -= ROBUST_EPSILON * vo_sign(robust_err);
        float x0 = xt0 * x_sign + (is_positive_slope ? 0.0f : -1.0f);

        /* a line of a PATH marker the layout does not count has no Path record: a robust (WebGPU) load reads zeros,
         * stride 0, no crossings (path_count.wgsl:112) */
        if (line.path_ix >= cfg->layout.n_paths) continue;
        vo_path path = paths[line.path_ix];
        int32_t bbox[4] = {(int32_t)path.bbox[0], (int32_t)path.bbox[1], (int32_t)path.bbox[2], (int32_t)path.bbox[3]};
        float xmin = vo_min(s0.x, s1.x);
        int32_t stride = bbox[2] - bbox[0];
        if (s0.y >= (float)bbox[3] || s1.y <= (float)bbox[1] || xmin >= (float)bbox[2] || stride == 0) continue;
        uint32_t imin_ = 0u;
        if (s0.y < (float)bbox[1]) {
            float iminf = vo_round(((float)bbox[1] - y0 + b - a) / (1.0f - a)) - 1.0f;
            if (y0 + iminf - floorf(a * iminf + b) < (float)bbox[1]) iminf += 1.0f;
            imin_ = f2u(iminf);
        }
        uint32_t imax_ = count;
        if (s1.y > (float)bbox[3]) {
            float imaxf = vo_round(((float)bbox[3] - y0 + b - a) / (1.0f - a)) - 1.0f;
            if (y0 + imaxf - floorf(a * imaxf + b) < (float)bbox[3]) imaxf += 1.0f;
            imax_ = f2u(imaxf);
        }
        int32_t delta = is_down ? -1 : 1;
        int32_t ymin = 0, ymax = 0;
        if (vo_max(s0.x, s1.x) <= (float)bbox[0]) {
            ymin = f2i(ceilf(s0.y));
            ymax = f2i(ceilf(s1.y));
            imax_ = imin_;
        } else {
            float fudge = is_positive_slope ? 0.0f : 1.0f;
            if (xmin < (float)bbox[0]) {
                float f = vo_round((x_sign * ((float)bbox[0] - x0) - b + fudge) / a);
                if ((x0 + x_sign * floorf(a * f + b) < (float)bbox[0]) == is_positive_slope) f += 1.0f;
                int32_t ynext = f2i(y0 + f - floorf(a * f + b) + 1.0f);
                if (is_positive_slope) {
                    if (f2u(f) > imin_) {
                        ymin = f2i(y0 + (y0 == s0.y ? 0.0f : 1.0f));
                        ymax = ynext;
                        imin_ = f2u(f);
                    }
                } else {
                    if (f2u(f) < imax_) {
                        ymin = ynext;
                        ymax = f2i(ceilf(s1.y));
                        imax_ = f2u(f);
                    }
                }
            }
            if (vo_max(s0.x, s1.x) > (float)bbox[2]) {
                float f = vo_round((x_sign * ((float)bbox[2] - x0) - b + fudge) / a);
                if ((x0 + x_sign * floorf(a * f + b) < (float)bbox[2]) == is_positive_slope) f += 1.0f;
                if (is_positive_slope) imax_ = umin(imax_, f2u(f));
                else imin_ = umax(imin_, f2u(f));
            }
        }
        imax_ = umax(imin_, imax_);
        ymin = imax(ymin, bbox[1]);
        ymax = imin(ymax, bbox[3]);
        for (int32_t y = ymin; y < ymax; y++) {
            int32_t base = (int32_t)path.tiles + (y - bbox[1]) * stride;
            if ((uint32_t)base < cfg->tiles_size) __atomic_fetch_add(&tile[base].backdrop, delta, __ATOMIC_RELAXED); /* robust access: an out-of-range store is dropped */
        }
        float last_z = floorf(a * ((float)imin_ - 1.0f) + b);
        uint32_t seg_base = __atomic_fetch_add(&bump->seg_counts, imax_ - imin_, __ATOMIC_RELAXED);
        for (uint32_t i = imin_; i < imax_; i++) {
            float zf = a * (float)i + b;
            float z = floorf(zf);
            int32_t y = f2i(y0 + (float)i - z);
            int32_t x = f2i(x0 + x_sign * z);
            int32_t base = (int32_t)path.tiles + (y - bbox[1]) * stride - bbox[0];
            int top_edge = (i == 0u) ? (y0 == s0.y) : (last_z == z);
            if (top_edge && x + 1 < bbox[2]) {
                int32_t x_bump = imax(x + 1, bbox[0]);
                if ((uint32_t)(base + x_bump) < cfg->tiles_size) __atomic_fetch_add(&tile[base + x_bump].backdrop, delta, __ATOMIC_RELAXED);
            }
            uint32_t seg_within_slice = 0u; /* robust access: an out-of-range atomic returns 0 and stores nothing */
            if ((uint32_t)(base + x) < cfg->tiles_size) {
                seg_within_slice = __atomic_fetch_add(&tile[base + x].segment_count_or_ix, 1u, __ATOMIC_RELAXED);
            }
            uint32_t seg_ix = seg_base + i - imin_;
            if (seg_ix < cfg->seg_counts_size) {
                seg_counts[seg_ix].line_ix = line_ix;
                seg_counts[seg_ix].counts = (seg_within_slice << 16) | i;
            }
            last_z = z;
        }
    }
}

/* ------------------------------------------------------------------ */
/* backdrop_dyn: shader/backdrop_dyn.wgsl:28-86, cpu/backdrop.rs:8-23  */
/* ------------------------------------------------------------------ */
void vo_stage_backdrop(vo_ctx *c) {
    const vo_config *cfg = &c->cfg;
    const vo_bump *bump = (const vo_bump *)c->buf[VO_BUF_BUMP];
    if (bump->failed != 0u) return;
    const vo_path *paths = (const vo_path *)c->buf[VO_BUF_PATHS];
    vo_tile *tiles = (vo_tile *)c->buf[VO_BUF_TILES];
    for (uint32_t d = 0; d < cfg->layout.n_draw_objects; d++) {
        vo_path path = paths[d];
        uint32_t width = path.bbox[2] - path.bbox[0];
        uint32_t height = path.bbox[3] - path.bbox[1];
        for (uint32_t y = 0; y < height; y++) {
            int32_t sum = 0;
            for (uint32_t x = 0; x < width; x++) {
                vo_tile *t = &tiles[path.tiles + y * width + x];
                sum += t->backdrop;
                t->backdrop = sum;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* coarse: shader/coarse.wgsl:62-471 (WGSL semantics: fixed 256-word   */
/* chunks + failure flag), cpu/coarse.rs for the sequential structure  */
/* ------------------------------------------------------------------ */
typedef struct {
    vo_ctx *c;
    vo_bump *bump;
    uint32_t *ptcl;
    uint32_t cmd_offset, cmd_limit;
} tile_state;

static void ptcl_write(tile_state *ts, uint32_t ix, uint32_t v) {
    if (ix < ts->c->cfg.ptcl_size) ts->ptcl[ix] = v;
}

static void alloc_cmd(tile_state *ts, uint32_t size) {
    if (ts->cmd_offset + size >= ts->cmd_limit) {
        const vo_config *cfg = &ts->c->cfg;
        uint32_t ptcl_dyn_start = cfg->width_in_tiles * cfg->height_in_tiles * PTCL_INITIAL_ALLOC;
        uint32_t new_cmd = ptcl_dyn_start + __atomic_fetch_add(&ts->bump->ptcl, PTCL_INCREMENT, __ATOMIC_RELAXED);
        if (new_cmd + PTCL_INCREMENT > cfg->ptcl_size) {
            new_cmd = 0u;
            __atomic_fetch_or(&ts->bump->failed, STAGE_COARSE, __ATOMIC_RELAXED);
        }
        ptcl_write(ts, ts->cmd_offset, CMD_JUMP);
        ptcl_write(ts, ts->cmd_offset + 1u, new_cmd);
        ts->cmd_offset = new_cmd;
        ts->cmd_limit = new_cmd + (PTCL_INCREMENT - PTCL_HEADROOM);
    }
}

static void write_path(tile_state *ts, vo_tile *tile, uint32_t draw_flags) {
    uint32_t n_segs = tile->segment_count_or_ix;
    if (n_segs != 0u) {
        uint32_t seg_ix = __atomic_fetch_add(&ts->bump->segments, n_segs, __ATOMIC_RELAXED);
        tile->segment_count_or_ix = ~seg_ix;
        alloc_cmd(ts, 4u);
        ptcl_write(ts, ts->cmd_offset, CMD_FILL);
        uint32_t even_odd = (draw_flags & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u;
        ptcl_write(ts, ts->cmd_offset + 1u, (n_segs << 1) | even_odd);
        ptcl_write(ts, ts->cmd_offset + 2u, seg_ix);
        ptcl_write(ts, ts->cmd_offset + 3u, (uint32_t)tile->backdrop);
        ts->cmd_offset += 4u;
    } else {
        alloc_cmd(ts, 1u);
        ptcl_write(ts, ts->cmd_offset, CMD_SOLID);
        ts->cmd_offset += 1u;
    }
}

static void write2(tile_state *ts, uint32_t a, uint32_t b) {
    alloc_cmd(ts, 2u);
    ptcl_write(ts, ts->cmd_offset, a);
    ptcl_write(ts, ts->cmd_offset + 1u, b);
    ts->cmd_offset += 2u;
}
static void write3(tile_state *ts, uint32_t a, uint32_t b, uint32_t cc) {
    alloc_cmd(ts, 3u);
    ptcl_write(ts, ts->cmd_offset, a);
    ptcl_write(ts, ts->cmd_offset + 1u, b);
    ptcl_write(ts, ts->cmd_offset + 2u, cc);
    ts->cmd_offset += 3u;
}

void vo_stage_coarse(vo_ctx *c) {
    const vo_config *cfg = &c->cfg;
    vo_bump *bump = (vo_bump *)c->buf[VO_BUF_BUMP];
    { /* coarse.wgsl:161-176 */
        uint32_t failed = bump->failed & (STAGE_BINNING | STAGE_TILE_ALLOC | STAGE_FLATTEN);
        if (bump->seg_counts > cfg->seg_counts_size) failed |= STAGE_PATH_COUNT;
        if (failed != 0u) {
            bump->failed |= failed;
            return;
        }
    }
    const uint32_t *scene = c->scene;
    const vo_draw_monoid *draw_monoids = (const vo_draw_monoid *)c->buf[VO_BUF_DRAW_MONOIDS];
    const vo_bin_header *bin_headers = (const vo_bin_header *)c->buf[VO_BUF_BIN_HEADERS];
    const uint32_t *info_bin_data = (const uint32_t *)c->buf[VO_BUF_INFO_BIN_DATA];
    const vo_path *paths = (const vo_path *)c->buf[VO_BUF_PATHS];
    vo_tile *tiles = (vo_tile *)c->buf[VO_BUF_TILES];
    uint32_t width_in_tiles = cfg->width_in_tiles, height_in_tiles = cfg->height_in_tiles;
    uint32_t width_in_bins = (width_in_tiles + N_TILE_X - 1u) / N_TILE_X;
    uint32_t height_in_bins = (height_in_tiles + N_TILE_Y - 1u) / N_TILE_Y;
    uint32_t n_bins = width_in_bins * height_in_bins;
    uint32_t aligned_n_bins = (n_bins + N_TILE - 1u) & ~(N_TILE - 1u);
    uint32_t drawtag_base = cfg->layout.draw_tag_base;
    uint32_t n_partitions = (cfg->layout.n_draw_objects + N_TILE - 1u) / N_TILE;

    /* one workgroup per bin, as the shader; the CPU-baseline mode runs the bins on n_threads threads (segment slices,
     * PTCL chunks and blend spill are bump atomics in the shader too) */
#pragma omp parallel if (c->n_threads > 1) num_threads(VO_OMP_THREADS(c))
    {
    /* per-tile compacted draw object lists for the current bin */
    uint32_t *list[N_TILE];
    uint32_t list_len[N_TILE], list_cap[N_TILE];
    for (uint32_t i = 0; i < N_TILE; i++) { list[i] = NULL; list_len[i] = 0; list_cap[i] = 0; }

#pragma omp for schedule(dynamic, 1)
    for (uint32_t bin = 0; bin < n_bins; bin++) {
        for (uint32_t i = 0; i < N_TILE; i++) list_len[i] = 0;
        uint32_t bin_x = bin % width_in_bins, bin_y = bin / width_in_bins;
        uint32_t bin_tile_x = N_TILE_X * bin_x, bin_tile_y = N_TILE_Y * bin_y;
        for (uint32_t part = 0; part < n_partitions; part++) {
            vo_bin_header bh = bin_headers[part * aligned_n_bins + bin];
            uint32_t start = cfg->layout.bin_data_start + bh.chunk_offset;
            for (uint32_t i = 0; i < bh.element_count; i++) {
                uint32_t drawobj_ix = info_bin_data[start + i];
                uint32_t tag = scene[drawtag_base + drawobj_ix];
                if (tag == DRAWTAG_NOP) continue;
                vo_draw_monoid dm = draw_monoids[drawobj_ix];
                vo_path path = paths[dm.path_ix];
                int32_t dx = (int32_t)path.bbox[0] - (int32_t)bin_tile_x;
                int32_t dy = (int32_t)path.bbox[1] - (int32_t)bin_tile_y;
                int32_t x0 = iclamp(dx, 0, (int32_t)N_TILE_X);
                int32_t y0 = iclamp(dy, 0, (int32_t)N_TILE_Y);
                int32_t x1 = iclamp((int32_t)path.bbox[2] - (int32_t)bin_tile_x, 0, (int32_t)N_TILE_X);
                int32_t y1 = iclamp((int32_t)path.bbox[3] - (int32_t)bin_tile_y, 0, (int32_t)N_TILE_Y);
                uint32_t stride = path.bbox[2] - path.bbox[0];
                /* coarse.wgsl:313-345: include_tile filter is evaluated before the clip state machine */
                uint32_t is_clip = (tag & 1u) != 0u;
                int is_blend = 0;
                if (is_clip) {
                    const uint32_t BLEND_CLIP = (128u << 8) | 3u;
                    uint32_t blend = scene[cfg->layout.draw_data_base + dm.scene_offset];
                    is_blend = blend != BLEND_CLIP;
                }
                uint32_t draw_flags = info_bin_data[dm.info_offset];
                int even_odd = (draw_flags & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u;
                for (int32_t y = y0; y < y1; y++) {
                    for (int32_t x = x0; x < x1; x++) {
                        uint32_t tile_ix = path.tiles + (uint32_t)((int32_t)stride * (y - dy) + (x - dx));
                        vo_tile t = tiles[tile_ix];
                        uint32_t n_segs = t.segment_count_or_ix;
                        int32_t bd = even_odd ? (abs(t.backdrop) & 1) : t.backdrop;
                        int backdrop_clear = bd == 0;
                        int include_tile = n_segs != 0u || (backdrop_clear == (int)is_clip) || is_blend;
                        if (!include_tile) continue;
                        uint32_t ti = (uint32_t)(y * (int32_t)N_TILE_X + x);
                        if (list_len[ti] == list_cap[ti]) {
                            list_cap[ti] = list_cap[ti] ? list_cap[ti] * 2 : 64;
                            list[ti] = (uint32_t *)realloc(list[ti], sizeof(uint32_t) * list_cap[ti]);
                        }
                        list[ti][list_len[ti]++] = drawobj_ix;
                    }
                }
            }
        }
        for (uint32_t tile_ix_in_bin = 0; tile_ix_in_bin < N_TILE; tile_ix_in_bin++) {
            uint32_t tile_x = tile_ix_in_bin % N_TILE_X, tile_y = tile_ix_in_bin / N_TILE_X;
            uint32_t this_tile_ix = (bin_tile_y + tile_y) * width_in_tiles + bin_tile_x + tile_x;
            tile_state ts;
            ts.c = c; ts.bump = bump; ts.ptcl = (uint32_t *)c->buf[VO_BUF_PTCL];
            ts.cmd_offset = this_tile_ix * PTCL_INITIAL_ALLOC;
            ts.cmd_limit = ts.cmd_offset + (PTCL_INITIAL_ALLOC - PTCL_HEADROOM);
            uint32_t blend_offset = ts.cmd_offset;
            ts.cmd_offset += 1u;
            uint32_t clip_depth = 0, clip_zero_depth = 0, render_blend_depth = 0, max_blend_depth = 0;
            int in_target = bin_tile_x + tile_x < width_in_tiles && bin_tile_y + tile_y < height_in_tiles;
            for (uint32_t li = 0; li < list_len[tile_ix_in_bin]; li++) {
                uint32_t drawobj_ix = list[tile_ix_in_bin][li];
                uint32_t drawtag = scene[drawtag_base + drawobj_ix];
                vo_draw_monoid dm = draw_monoids[drawobj_ix];
                uint32_t dd = cfg->layout.draw_data_base + dm.scene_offset;
                uint32_t di = dm.info_offset;
                uint32_t draw_flags = info_bin_data[di];
                if (clip_zero_depth == 0u) {
                    vo_path path = paths[dm.path_ix];
                    uint32_t stride = path.bbox[2] - path.bbox[0];
                    uint32_t x = bin_tile_x + tile_x - path.bbox[0];
                    uint32_t y = bin_tile_y + tile_y - path.bbox[1];
                    vo_tile *tile = &tiles[path.tiles + y * stride + x];
                    switch (drawtag) {
                    case DRAWTAG_FILL_COLOR:
                        write_path(&ts, tile, draw_flags);
                        write2(&ts, CMD_COLOR, scene[dd]);
                        break;
                    case DRAWTAG_BLURRED_ROUNDED_RECT:
                        write_path(&ts, tile, draw_flags);
                        write3(&ts, CMD_BLUR_RECT, di + 1u, scene[dd]);
                        break;
                    case DRAWTAG_FILL_LIN_GRADIENT:
                        write_path(&ts, tile, draw_flags);
                        write3(&ts, CMD_LIN_GRAD, scene[dd], di + 1u);
                        break;
                    case DRAWTAG_FILL_RAD_GRADIENT:
                        write_path(&ts, tile, draw_flags);
                        write3(&ts, CMD_RAD_GRAD, scene[dd], di + 1u);
                        break;
                    case DRAWTAG_FILL_SWEEP_GRADIENT:
                        write_path(&ts, tile, draw_flags);
                        write3(&ts, CMD_SWEEP_GRAD, scene[dd], di + 1u);
                        break;
                    case DRAWTAG_FILL_IMAGE:
                        write_path(&ts, tile, draw_flags);
                        write2(&ts, CMD_IMAGE, di + 1u);
                        break;
                    case DRAWTAG_BEGIN_CLIP: {
                        int even_odd = (draw_flags & DRAW_INFO_FLAGS_FILL_RULE_BIT) != 0u;
                        int32_t bd = even_odd ? (abs(tile->backdrop) & 1) : tile->backdrop;
                        if (tile->segment_count_or_ix == 0u && bd == 0) {
                            clip_zero_depth = clip_depth + 1u;
                        } else {
                            alloc_cmd(&ts, 1u);
                            ptcl_write(&ts, ts.cmd_offset, CMD_BEGIN_CLIP);
                            ts.cmd_offset += 1u;
                            render_blend_depth += 1u;
                            max_blend_depth = umax(max_blend_depth, render_blend_depth);
                        }
                        clip_depth += 1u;
                        break;
                    }
                    case DRAWTAG_END_CLIP:
                        clip_depth -= 1u;
                        write_path(&ts, tile, draw_flags);
                        write3(&ts, CMD_END_CLIP, scene[dd], scene[dd + 1u]);
                        render_blend_depth -= 1u;
                        break;
                    default: break;
                    }
                } else {
                    if (drawtag == DRAWTAG_BEGIN_CLIP) {
                        clip_depth += 1u;
                    } else if (drawtag == DRAWTAG_END_CLIP) {
                        if (clip_depth == clip_zero_depth) clip_zero_depth = 0u;
                        clip_depth -= 1u;
                    }
                }
            }
            if (in_target) {
                ptcl_write(&ts, ts.cmd_offset, CMD_END);
                uint32_t blend_ix = 0u;
                if (max_blend_depth > BLEND_STACK_SPLIT) {
                    uint32_t scratch_size = (max_blend_depth - BLEND_STACK_SPLIT) * TILE_WIDTH * TILE_HEIGHT;
                    blend_ix = __atomic_fetch_add(&bump->blend, scratch_size, __ATOMIC_RELAXED);
                    if (blend_ix + scratch_size > cfg->blend_size) __atomic_fetch_or(&bump->failed, STAGE_COARSE, __ATOMIC_RELAXED);
                }
                ptcl_write(&ts, blend_offset, blend_ix);
            }
        }
    }
    for (uint32_t i = 0; i < N_TILE; i++) free(list[i]);
    }
}

/* ------------------------------------------------------------------ */
/* path_tiling_setup + path_tiling: shader/path_tiling_setup.wgsl,     */
/* shader/path_tiling.wgsl:39-173, cpu/path_tiling.rs                  */
/* ------------------------------------------------------------------ */
void vo_stage_path_tiling(vo_ctx *c) {
    const vo_config *cfg = &c->cfg;
    vo_bump *bump = (vo_bump *)c->buf[VO_BUF_BUMP];
    uint32_t *ptcl = (uint32_t *)c->buf[VO_BUF_PTCL];
    if (bump->failed != 0u) { /* path_tiling_setup.wgsl:21-25 */
        ptcl[0] = ~0u;
        return;
    }
    const vo_seg_count *seg_counts = (const vo_seg_count *)c->buf[VO_BUF_SEG_COUNTS];
    const vo_line_soup *lines = (const vo_line_soup *)c->buf[VO_BUF_LINES];
    const vo_path *paths = (const vo_path *)c->buf[VO_BUF_PATHS];
    const vo_tile *tiles = (const vo_tile *)c->buf[VO_BUF_TILES];
    vo_segment *segments = (vo_segment *)c->buf[VO_BUF_SEGMENTS];
    const float TILE_SCALE = 0.0625f;
    uint32_t n_segments = bump->seg_counts;
#pragma omp parallel for schedule(static) if (c->n_threads > 1) num_threads(VO_OMP_THREADS(c))
    for (uint32_t gi = 0; gi < n_segments; gi++) {
        vo_seg_count sc = seg_counts[gi];
        vo_line_soup line = lines[sc.line_ix];
        uint32_t seg_within_slice = sc.counts >> 16;
        uint32_t seg_within_line = sc.counts & 0xffffu;
        int is_down = line.p1[1] >= line.p0[1];
        vec2 xy0 = is_down ? v2(line.p0[0], line.p0[1]) : v2(line.p1[0], line.p1[1]);
        vec2 xy1 = is_down ? v2(line.p1[0], line.p1[1]) : v2(line.p0[0], line.p0[1]);
        vec2 s0 = vmul(xy0, TILE_SCALE);
        vec2 s1 = vmul(xy1, TILE_SCALE);
        uint32_t count_x = vo_span(s0.x, s1.x) - 1u;
        uint32_t count = count_x + vo_span(s0.y, s1.y);
        float dx = fabsf(s1.x - s0.x);
        float dy = s1.y - s0.y;
        float idxdy = 1.0f / (dx + dy);
        float a = dx * idxdy;
        int is_positive_slope = s1.x >= s0.x;
        float x_sign = is_positive_slope ? 1.0f : -1.0f;
        float xt0 = floorf(s0.x * x_sign);
        float cc = s0.x * x_sign - xt0;
        float y0i = floorf(s0.y);
        float ytop = (s0.y == s1.y) ? ceilf(s0.y) : y0i + 1.0f;
        float b = vo_min((dy * cc + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
        float robust_err = floorf(a * ((float)count - 1.0f) + b) - (float)count_x;
        if (robust_err != 0.0f) a -= ROBUST_EPSILON * vo_sign(robust_err);
        int32_t x0i = f2i(xt0 * x_sign + 0.5f * (x_sign - 1.0f));
        float z = floorf(a * (float)seg_within_line + b);
        int32_t x = x0i + f2i(x_sign * z);
        int32_t y = f2i(y0i + (float)seg_within_line - z);

        vo_path path = paths[line.path_ix];
        int32_t bbox[4] = {(int32_t)path.bbox[0], (int32_t)path.bbox[1], (int32_t)path.bbox[2], (int32_t)path.bbox[3]};
        int32_t stride = bbox[2] - bbox[0];
        int32_t tile_ix = (int32_t)path.tiles + (y - bbox[1]) * stride + x - bbox[0];
        vo_tile tile = {0, 0u}; /* robust access: an out-of-range load reads zero (a crossing index past 16 bits, the
                                   reference's own limit in path_count.wgsl:196, recomputes a tile that is not the path's) */
        if ((uint32_t)tile_ix < cfg->tiles_size) tile = tiles[tile_ix];
        uint32_t seg_start = ~tile.segment_count_or_ix;
        if ((int32_t)seg_start < 0) continue;
        vec2 tile_xy = v2((float)x * (float)TILE_WIDTH, (float)y * (float)TILE_HEIGHT);
        vec2 tile_xy1 = v2(tile_xy.x + (float)TILE_WIDTH, tile_xy.y + (float)TILE_HEIGHT);
        if (seg_within_line > 0u) {
            float z_prev = floorf(a * ((float)seg_within_line - 1.0f) + b);
            if (z == z_prev) {
                float xt = xy0.x + (xy1.x - xy0.x) * (tile_xy.y - xy0.y) / (xy1.y - xy0.y);
                xt = vo_clamp(xt, tile_xy.x + 1e-3f, tile_xy1.x);
                xy0 = v2(xt, tile_xy.y);
            } else {
                float x_clip = is_positive_slope ? tile_xy.x : tile_xy1.x;
                float yt = xy0.y + (xy1.y - xy0.y) * (x_clip - xy0.x) / (xy1.x - xy0.x);
                yt = vo_clamp(yt, tile_xy.y + 1e-3f, tile_xy1.y);
                xy0 = v2(x_clip, yt);
            }
        }
        if (seg_within_line < count - 1u) {
            float z_next = floorf(a * ((float)seg_within_line + 1.0f) + b);
            if (z == z_next) {
                float xt = xy0.x + (xy1.x - xy0.x) * (tile_xy1.y - xy0.y) / (xy1.y - xy0.y);
                xt = vo_clamp(xt, tile_xy.x + 1e-3f, tile_xy1.x);
                xy1 = v2(xt, tile_xy1.y);
            } else {
                float x_clip = is_positive_slope ? tile_xy1.x : tile_xy.x;
                float yt = xy0.y + (xy1.y - xy0.y) * (x_clip - xy0.x) / (xy1.x - xy0.x);
                yt = vo_clamp(yt, tile_xy.y + 1e-3f, tile_xy1.y);
                xy1 = v2(x_clip, yt);
            }
        }
        float y_edge = 1e9f;
        vec2 p0 = vsub(xy0, tile_xy);
        vec2 p1 = vsub(xy1, tile_xy);
        const float EPSILON = 1e-6f;
        if (p0.x == 0.0f) {
            if (p1.x == 0.0f) {
                p0.x = EPSILON;
                if (p0.y == 0.0f) {
                    p1.x = EPSILON;
                    p1.y = (float)TILE_HEIGHT;
                } else {
                    p1.x = 2.0f * EPSILON;
                    p1.y = p0.y;
                }
            } else if (p0.y == 0.0f) {
                p0.x = EPSILON;
            } else {
                y_edge = p0.y;
            }
        } else if (p1.x == 0.0f) {
            if (p1.y == 0.0f) {
                p1.x = EPSILON;
            } else {
                y_edge = p1.y;
            }
        }
        if (p0.x == floorf(p0.x) && p0.x != 0.0f) p0.x -= EPSILON;
        if (p1.x == floorf(p1.x) && p1.x != 0.0f) p1.x -= EPSILON;
        if (!is_down) { vec2 tmp = p0; p0 = p1; p1 = tmp; }
        uint32_t out_ix = seg_start + seg_within_slice;
        if (out_ix < cfg->segments_size) {
            vo_segment *sg = &segments[out_ix];
            sg->p0[0] = p0.x; sg->p0[1] = p0.y;
            sg->p1[0] = p1.x; sg->p1[1] = p1.y;
            sg->y_edge = y_edge;
            sg->pad = 0;
        }
    }
}
