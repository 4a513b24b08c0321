/*
 * Oracle fine rasterizer: literal sequential restatement of
 * vello_shaders/shader/fine.wgsl (area :1005-1059, msaa :146-709,
 * interpreter :1064-1398) and shader/shared/blend.wgsl.
 * The 64 invocations of a workgroup are emulated phase by phase (a phase is
 * the code between two workgroupBarrier()s), on the same packed SWAR data.
 * TEST INFRASTRUCTURE ONLY (see vello_oracle.h).
 */
#include "vo_internal.h"
#include <pthread.h>

#define WG_SIZE 64u
#define PIXELS_PER_THREAD 4u
#define GRADIENT_WIDTH 512
#define LUMINANCE_MASK_LAYER 0x10000u

typedef struct { float v[4]; } vec4;

typedef struct {
    uint32_t sh_count[64];
    uint32_t sh_winding_y[4];
    uint32_t sh_winding_y_prefix[4];
    uint32_t sh_winding[64];
    uint32_t sh_samples[1024];
} fine_shared;

typedef struct { uint32_t size_and_rule, seg_data; int32_t backdrop; } cmd_fill;

static vec4 unpack4x8unorm(uint32_t u) {
    vec4 r;
    r.v[0] = (float)(u & 0xffu) / 255.0f;
    r.v[1] = (float)((u >> 8) & 0xffu) / 255.0f;
    r.v[2] = (float)((u >> 16) & 0xffu) / 255.0f;
    r.v[3] = (float)((u >> 24) & 0xffu) / 255.0f;
    return r;
}
/* WGSL pack4x8unorm: floor(0.5 + 255 * clamp(e, 0, 1)) */
static uint32_t unorm8(float e) { return (uint32_t)floorf(0.5f + 255.0f * vo_clamp(e, 0.0f, 1.0f)); }
static uint32_t pack4x8unorm(vec4 c) {
    return unorm8(c.v[0]) | (unorm8(c.v[1]) << 8) | (unorm8(c.v[2]) << 16) | (unorm8(c.v[3]) << 24);
}

/* ---------------- area AA: fine.wgsl:1005-1059 ---------------- */
static void fill_path_area(const vo_ctx *c, cmd_fill fill, float area[256]) {
    const vo_segment *segments = (const vo_segment *)c->buf[VO_BUF_SEGMENTS];
    uint32_t n_segs = fill.size_and_rule >> 1;
    int even_odd = (fill.size_and_rule & 1u) != 0u;
    float backdrop_f = (float)fill.backdrop;
    for (uint32_t th = 0; th < 64u; th++) {
        float xy_x = (float)((th & 3u) * PIXELS_PER_THREAD);
        float xy_y = (float)(th >> 2);
        float ar[4] = {backdrop_f, backdrop_f, backdrop_f, backdrop_f};
        for (uint32_t i = 0; i < n_segs; i++) {
            vo_segment sg = segments[fill.seg_data + i];
            float y = sg.p0[1] - xy_y;
            float delta_x = sg.p1[0] - sg.p0[0];
            float delta_y = sg.p1[1] - sg.p0[1];
            float y0 = vo_clamp(y, 0.0f, 1.0f);
            float y1 = vo_clamp(y + delta_y, 0.0f, 1.0f);
            float dy = y0 - y1;
            if (dy != 0.0f) {
                float vec_y_recip = 1.0f / delta_y;
                float t0 = (y0 - y) * vec_y_recip;
                float t1 = (y1 - y) * vec_y_recip;
                float startx = sg.p0[0] - xy_x;
                float x0 = startx + t0 * delta_x;
                float x1 = startx + t1 * delta_x;
                float xmin0 = vo_min(x0, x1);
                float xmax0 = vo_max(x0, x1);
                for (uint32_t k = 0; k < PIXELS_PER_THREAD; k++) {
                    float i_f = (float)k;
                    float xmin = vo_min(xmin0 - i_f, 1.0f) - 1.0e-6f;
                    float xmax = xmax0 - i_f;
                    float b = vo_min(xmax, 1.0f);
                    float cc = vo_max(b, 0.0f);
                    float d = vo_max(xmin, 0.0f);
                    float a = (b + 0.5f * (d * d - cc * cc) - xmin) / (xmax - xmin);
                    ar[k] += a * dy;
                }
            }
            float y_edge = vo_sign(delta_x) * vo_clamp(xy_y - sg.y_edge + 1.0f, 0.0f, 1.0f);
            for (uint32_t k = 0; k < PIXELS_PER_THREAD; k++) ar[k] += y_edge;
        }
        for (uint32_t k = 0; k < PIXELS_PER_THREAD; k++) {
            float a = ar[k];
            if (even_odd) a = fabsf(a - 2.0f * vo_round(0.5f * a));
            else a = vo_min(fabsf(a), 1.0f);
            area[th * 4u + k] = a;
        }
    }
}

/* ---------------- MSAA: fine.wgsl:146-709 ---------------- */
typedef struct {
    /* per-pixel DDA setup shared by nonzero / even-odd (fine.wgsl:236-261) */
    int is_down, is_positive_slope;
    float a, b, x_sign, y0i;
    int32_t x0i;
    vec2 xy0, xy1;
} line_setup;

static line_setup setup_line(vo_segment sg) {
    line_setup s;
    vec2 xy0_in = v2(sg.p0[0], sg.p0[1]), xy1_in = v2(sg.p1[0], sg.p1[1]);
    s.is_down = xy1_in.y >= xy0_in.y;
    s.xy0 = s.is_down ? xy0_in : xy1_in;
    s.xy1 = s.is_down ? xy1_in : xy0_in;
    float dx = fabsf(s.xy1.x - s.xy0.x);
    float dy = s.xy1.y - s.xy0.y;
    float idxdy = 1.0f / (dx + dy);
    float a = dx * idxdy;
    s.is_positive_slope = s.xy1.x >= s.xy0.x;
    s.x_sign = s.is_positive_slope ? 1.0f : -1.0f;
    float xt0 = floorf(s.xy0.x * s.x_sign);
    float cc = s.xy0.x * s.x_sign - xt0;
    s.y0i = floorf(s.xy0.y);
    float ytop = s.y0i + 1.0f;
    s.b = vo_min((dy * cc + dx * (ytop - s.xy0.y)) * idxdy, ONE_MINUS_ULP);
    uint32_t count_x = vo_span(s.xy0.x, s.xy1.x) - 1u;
    uint32_t count = count_x + vo_span(s.xy0.y, s.xy1.y);
    float robust_err = floorf(a * ((float)count - 1.0f) + s.b) - (float)count_x;
    if (robust_err != 0.0f) a -= ROBUST_EPSILON * vo_sign(robust_err);
    s.a = a;
    s.x0i = f2i(xt0 * s.x_sign + 0.5f * (s.x_sign - 1.0f));
    return s;
}

static uint32_t load_mask(const vo_ctx *c, int msaa16, uint32_t mask_ix) {
    if (msaa16) {
        const uint32_t *lut = (const uint32_t *)c->mask_lut16;
        return (lut[mask_ix / 2u] >> ((mask_ix % 2u) * 16u)) & 0xffffu;
    } else {
        const uint32_t *lut = (const uint32_t *)c->mask_lut8;
        return (lut[mask_ix / 4u] >> ((mask_ix % 4u) * 8u)) & 0xffu;
    }
}

static void fill_path_ms(const vo_ctx *c, fine_shared *sh, cmd_fill fill, int msaa16, float area[256]) {
    const vo_segment *segments = (const vo_segment *)c->buf[VO_BUF_SEGMENTS];
    const uint32_t MASK_WIDTH = msaa16 ? 64u : 32u, MASK_HEIGHT = msaa16 ? 64u : 32u;
    const uint32_t SWPP = msaa16 ? 4u : 2u; /* SAMPLE_WORDS_PER_PIXEL */
    const uint32_t nsamp = msaa16 ? 16u : 8u;
    const uint32_t full = msaa16 ? 0xffffu : 0xffu;
    int even_odd = (fill.size_and_rule & 1u) != 0u;
    uint32_t n_segs = fill.size_and_rule >> 1;
    /* init (fine.wgsl:157-169 / :514-523) */
    if (!even_odd) {
        for (uint32_t i = 0; i < 4; i++) sh->sh_winding_y[i] = 0x80808080u;
        for (uint32_t i = 0; i < 64; i++) sh->sh_winding[i] = 0x80808080u;
        for (uint32_t i = 0; i < 64u * PIXELS_PER_THREAD * SWPP; i++) sh->sh_samples[i] = 0x80808080u;
    } else {
        sh->sh_winding_y[0] = 0u;
        for (uint32_t i = 0; i < 16; i++) sh->sh_winding[i] = 0u;
        for (uint32_t i = 0; i < 256u; i++) sh->sh_samples[i] = 0u;
    }
    uint32_t n_batch = (n_segs + (WG_SIZE - 1u)) / WG_SIZE;
    for (uint32_t batch = 0; batch < n_batch; batch++) {
        uint32_t slice_size = umin(n_segs - batch * WG_SIZE, WG_SIZE);
        /* count phase */
        for (uint32_t th = 0; th < WG_SIZE; th++) {
            uint32_t count = 0u;
            if (th < slice_size) {
                vo_segment sg = segments[fill.seg_data + batch * WG_SIZE + th];
                vec2 xy0 = v2(sg.p0[0], sg.p0[1]), xy1 = v2(sg.p1[0], sg.p1[1]);
                float y_edge_f = (float)TILE_HEIGHT;
                int32_t delta = (xy1.x <= xy0.x) ? 1 : -1;
                if (xy0.x == 0.0f) y_edge_f = xy0.y;
                else if (xy1.x == 0.0f) y_edge_f = xy1.y;
                if (!(xy0.y == xy1.y && xy0.y == floorf(xy0.y)))
                    count = vo_span(xy0.x, xy1.x) + vo_span(xy0.y, xy1.y) - 1u;
                uint32_t y_edge = f2u(ceilf(y_edge_f));
                if (y_edge < TILE_HEIGHT) {
                    if (!even_odd) sh->sh_winding_y[y_edge >> 2] += ((uint32_t)delta) << ((y_edge & 3u) << 3);
                    else sh->sh_winding_y[0] ^= 1u << y_edge;
                }
            }
            sh->sh_count[th] = count;
        }
        /* inclusive prefix over the slice */
        for (uint32_t th = 1; th < slice_size; th++) sh->sh_count[th] += sh->sh_count[th - 1u];
        uint32_t total = sh->sh_count[slice_size - 1u];
        /* pixel phase */
        for (uint32_t i = 0; i < total; i++) {
            uint32_t lo = 0u, hi = slice_size;
            while (hi > lo + 1u) {
                uint32_t mid = (lo + hi) >> 1;
                if (i >= sh->sh_count[mid - 1u]) lo = mid; else hi = mid;
            }
            uint32_t el_ix = lo;
            int last_pixel = i + 1u == sh->sh_count[el_ix];
            uint32_t sub_ix = i - (el_ix > 0u ? sh->sh_count[el_ix - 1u] : 0u);
            vo_segment sg = segments[fill.seg_data + batch * WG_SIZE + el_ix];
            line_setup s = setup_line(sg);
            float zf = s.a * (float)sub_ix + s.b;
            float z = floorf(zf);
            int32_t x = s.x0i + f2i(s.x_sign * z);
            int32_t y = f2i(s.y0i) + (int32_t)sub_ix - f2i(z);
            int is_delta, is_bump = 0;
            float zp = floorf(s.a * (float)(sub_ix - 1u) + s.b);
            if (sub_ix == 0u) {
                is_delta = s.y0i == s.xy0.y;
                if (!even_odd) is_bump = s.xy0.x == 0.0f && s.y0i != s.xy0.y;
                else is_bump = s.xy0.x == 0.0f;
            } else {
                is_delta = z == zp;
                is_bump = s.is_positive_slope && !is_delta;
            }
            uint32_t pix_ix = (uint32_t)y * TILE_WIDTH + (uint32_t)x;
            if ((uint32_t)x < TILE_WIDTH - 1u && (uint32_t)y < TILE_HEIGHT) {
                if (is_delta) {
                    if (!even_odd) {
                        uint32_t delta_pix = pix_ix + 1u;
                        uint32_t d = (s.is_down ? 1u : 0xffffffffu) << ((delta_pix & 3u) << 3);
                        sh->sh_winding[delta_pix >> 2] += d;
                    } else {
                        sh->sh_winding[y] ^= 2u << (uint32_t)x;
                    }
                }
            }
            uint32_t mask_block = (uint32_t)s.is_positive_slope * (MASK_WIDTH * MASK_HEIGHT / 2u);
            float half_height = (float)(MASK_HEIGHT / 2u);
            float mask_row = floorf(vo_min(s.a * half_height, half_height - 1.0f)) * (float)MASK_WIDTH;
            float mask_col = floorf((zf - z) * (float)MASK_WIDTH);
            uint32_t mask_ix = mask_block + f2u(mask_row + mask_col);
            uint32_t mask = load_mask(c, msaa16, mask_ix);
            if (sub_ix == 0u && !is_bump) {
                uint32_t mask_shift = f2u(vo_round((float)nsamp * (s.xy0.y - (float)y)));
                mask &= mask_shift < 32u ? (full << mask_shift) : 0u;
            }
            if (last_pixel && s.xy1.x != 0.0f) {
                uint32_t mask_shift = f2u(vo_round((float)nsamp * (s.xy1.y - (float)y)));
                mask &= ~(mask_shift < 32u ? (full << mask_shift) : 0u);
            }
            if (pix_ix >= 256u) continue; /* memory safety; unreachable for tile-clipped segments */
            if (even_odd) {
                if (is_bump) mask ^= full;
                sh->sh_samples[pix_ix] ^= mask;
                continue;
            }
            uint32_t bump_delta = s.is_down ? 0x1010101u : (uint32_t)(-0x1010101);
            uint32_t nwords = msaa16 ? 2u : 1u; /* 8-bit halves */
            for (uint32_t h = 0; h < nwords; h++) {
                uint32_t m8 = (mask >> (8u * h)) & 0xffu;
                uint32_t m_a = m8 ^ (m8 << 7);
                uint32_t m_b = m_a ^ (m_a << 14);
                uint32_t e0 = m_b & 0x1010101u;
                uint32_t s0 = s.is_down ? (uint32_t)(-(int32_t)e0) : e0;
                uint32_t e1 = (m_b >> 4) & 0x1010101u;
                uint32_t s1 = s.is_down ? (uint32_t)(-(int32_t)e1) : e1;
                if (is_bump) { s0 += bump_delta; s1 += bump_delta; }
                sh->sh_samples[pix_ix * SWPP + 2u * h] += s0;
                sh->sh_samples[pix_ix * SWPP + 2u * h + 1u] += s1;
            }
        }
    }
    /* resolve */
    if (even_odd) {
        uint32_t scan_y = sh->sh_winding_y[0];
        scan_y ^= scan_y << 1; scan_y ^= scan_y << 2; scan_y ^= scan_y << 4; scan_y ^= scan_y << 8;
        for (uint32_t th = 0; th < WG_SIZE; th++) {
            uint32_t ly = th >> 2;
            uint32_t scan_x = sh->sh_winding[ly];
            scan_x ^= scan_x << 1; scan_x ^= scan_x << 2; scan_x ^= scan_x << 4; scan_x ^= scan_x << 8;
            uint32_t row_parity = (scan_y >> ly) ^ (uint32_t)fill.backdrop;
            for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) {
                uint32_t pix_ix = th * PIXELS_PER_THREAD + i;
                uint32_t samples = sh->sh_samples[pix_ix];
                uint32_t pix_parity = row_parity ^ (scan_x >> (pix_ix % TILE_WIDTH));
                uint32_t pix_mask = (uint32_t)(-(int32_t)(pix_parity & 1u));
                area[pix_ix] = (float)popcnt((samples ^ pix_mask) & full) * (msaa16 ? 0.0625f : 0.125f);
            }
        }
        return;
    }
    uint32_t packed_w_arr[64], wind_y_arr[64];
    uint32_t new_winding[64];
    for (uint32_t th = 0; th < WG_SIZE; th++) {
        uint32_t lx = th & 3u, ly = th >> 2;
        uint32_t major = th; /* (th*4)>>2 */
        uint32_t packed_w = sh->sh_winding[major];
        packed_w += (packed_w - 0x808080u) << 8;
        packed_w += (packed_w - 0x8080u) << 16;
        uint32_t packed_y = sh->sh_winding_y[ly >> 2];
        packed_y += (packed_y - 0x808080u) << 8;
        packed_y += (packed_y - 0x8080u) << 16;
        uint32_t wind_y = (packed_y >> ((ly & 3u) << 3)) - 0x80u;
        if ((ly & 3u) == 3u && lx == 0u) sh->sh_winding_y_prefix[ly >> 2] = wind_y;
        uint32_t prefix_x = ((packed_w >> 24) - 0x80u) * 0x1010101u;
        new_winding[major] = prefix_x;
        packed_w_arr[th] = packed_w;
        wind_y_arr[th] = wind_y;
    }
    for (uint32_t i = 0; i < 64; i++) sh->sh_winding[i] = new_winding[i];
    for (uint32_t th = 0; th < WG_SIZE; th++) {
        uint32_t ly = th >> 2;
        uint32_t major = th;
        uint32_t packed_w = packed_w_arr[th];
        uint32_t wind_y = wind_y_arr[th];
        for (uint32_t i = (major & ~3u); i < major; i++) packed_w += sh->sh_winding[i];
        for (uint32_t i = 0; i < (ly >> 2); i++) wind_y += sh->sh_winding_y_prefix[i];
        for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) {
            uint32_t pix_ix = th * PIXELS_PER_THREAD + i;
            uint32_t expected_zero = (((packed_w >> (i * 8u)) + wind_y) & 0xffu) - (uint32_t)fill.backdrop;
            if (expected_zero >= 256u) {
                area[pix_ix] = 1.0f;
            } else if (!msaa16) {
                uint32_t samples0 = sh->sh_samples[pix_ix * 2u];
                uint32_t samples1 = sh->sh_samples[pix_ix * 2u + 1u];
                uint32_t xored0 = (expected_zero * 0x1010101u) ^ samples0;
                uint32_t xored0_2 = xored0 | (xored0 * 2u);
                uint32_t xored1 = (expected_zero * 0x1010101u) ^ samples1;
                uint32_t xored1_2 = xored1 | (xored1 >> 1);
                uint32_t xored2 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
                uint32_t xored4 = xored2 | (xored2 * 4u);
                uint32_t xored8 = xored4 | (xored4 * 16u);
                area[pix_ix] = (float)popcnt(xored8 & 0xC0C0C0C0u) * 0.125f;
            } else {
                uint32_t samples0 = sh->sh_samples[pix_ix * 4u];
                uint32_t samples1 = sh->sh_samples[pix_ix * 4u + 1u];
                uint32_t samples2 = sh->sh_samples[pix_ix * 4u + 2u];
                uint32_t samples3 = sh->sh_samples[pix_ix * 4u + 3u];
                uint32_t xored0 = (expected_zero * 0x1010101u) ^ samples0;
                uint32_t xored0_2 = xored0 | (xored0 * 2u);
                uint32_t xored1 = (expected_zero * 0x1010101u) ^ samples1;
                uint32_t xored1_2 = xored1 | (xored1 >> 1);
                uint32_t xored01 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
                uint32_t xored01_4 = xored01 | (xored01 * 4u);
                uint32_t xored2 = (expected_zero * 0x1010101u) ^ samples2;
                uint32_t xored2_2 = xored2 | (xored2 * 2u);
                uint32_t xored3 = (expected_zero * 0x1010101u) ^ samples3;
                uint32_t xored3_2 = xored3 | (xored3 >> 1);
                uint32_t xored23 = (xored2_2 & 0xAAAAAAAAu) | (xored3_2 & 0x55555555u);
                uint32_t xored23_4 = xored23 | (xored23 >> 2);
                uint32_t xored4 = (xored01_4 & 0xCCCCCCCCu) | (xored23_4 & 0x33333333u);
                uint32_t xored8 = xored4 | (xored4 * 16u);
                area[pix_ix] = (float)popcnt(xored8 & 0xF0F0F0F0u) * 0.0625f;
            }
        }
    }
}

/* ---------------- blend: shader/shared/blend.wgsl ---------------- */
typedef struct { float x, y, z; } vec3;
static vec3 v3(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static float min3(vec3 c) { return vo_min(c.x, vo_min(c.y, c.z)); }
static float max3(vec3 c) { return vo_max(c.x, vo_max(c.y, c.z)); }
static float lum(vec3 c) { return c.x * 0.3f + c.y * 0.59f + c.z * 0.11f; }
static float svg_lum(vec3 c) { return c.x * 0.2125f + c.y * 0.7154f + c.z * 0.0721f; }
static float sat(vec3 c) { return max3(c) - min3(c); }
static float screen1(float cb, float cs) { return cb + cs - (cb * cs); }
static float color_dodge(float cb, float cs) {
    if (cb == 0.0f) return 0.0f;
    else if (cs == 1.0f) return 1.0f;
    else return vo_min(1.0f, cb / (1.0f - cs));
}
static float color_burn(float cb, float cs) {
    if (cb == 1.0f) return 1.0f;
    else if (cs == 0.0f) return 0.0f;
    else return 1.0f - vo_min(1.0f, (1.0f - cb) / cs);
}
static float hard_light1(float cb, float cs) {
    return cs <= 0.5f ? cb * 2.0f * cs : screen1(cb, 2.0f * cs - 1.0f);
}
static float soft_light1(float cb, float cs) {
    float d = cb <= 0.25f ? ((16.0f * cb - 12.0f) * cb + 4.0f) * cb : sqrtf(cb);
    return cs <= 0.5f ? cb - (1.0f - 2.0f * cs) * cb * (1.0f - cb) : cb + (2.0f * cs - 1.0f) * (d - cb);
}
static vec3 clip_color(vec3 c) {
    float l = lum(c);
    float n = min3(c);
    float x = max3(c);
    if (n < 0.0f) {
        c = v3(l + (((c.x - l) * l) / (l - n)), l + (((c.y - l) * l) / (l - n)), l + (((c.z - l) * l) / (l - n)));
    }
    if (x > 1.0f) {
        c = v3(l + (((c.x - l) * (1.0f - l)) / (x - l)), l + (((c.y - l) * (1.0f - l)) / (x - l)),
               l + (((c.z - l) * (1.0f - l)) / (x - l)));
    }
    return c;
}
static vec3 set_lum(vec3 c, float l) {
    float d = l - lum(c);
    return clip_color(v3(c.x + d, c.y + d, c.z + d));
}
static void set_sat_inner(float *cmin, float *cmid, float *cmax, float s) {
    if (*cmax > *cmin) {
        *cmid = ((*cmid - *cmin) * s) / (*cmax - *cmin);
        *cmax = s;
    } else {
        *cmid = 0.0f;
        *cmax = 0.0f;
    }
    *cmin = 0.0f;
}
static vec3 set_sat(vec3 c, float s) {
    float r = c.x, g = c.y, b = c.z;
    if (r <= g) {
        if (g <= b) set_sat_inner(&r, &g, &b, s);
        else if (r <= b) set_sat_inner(&r, &b, &g, s);
        else set_sat_inner(&b, &r, &g, s);
    } else {
        if (r <= b) set_sat_inner(&g, &r, &b, s);
        else if (g <= b) set_sat_inner(&g, &b, &r, s);
        else set_sat_inner(&b, &g, &r, s);
    }
    return v3(r, g, b);
}
static vec3 blend_mix(vec3 cb, vec3 cs, uint32_t mode) {
    switch (mode) {
    case 1: return v3(cb.x * cs.x, cb.y * cs.y, cb.z * cs.z);
    case 2: return v3(screen1(cb.x, cs.x), screen1(cb.y, cs.y), screen1(cb.z, cs.z));
    case 3: return v3(hard_light1(cs.x, cb.x), hard_light1(cs.y, cb.y), hard_light1(cs.z, cb.z));
    case 4: return v3(vo_min(cb.x, cs.x), vo_min(cb.y, cs.y), vo_min(cb.z, cs.z));
    case 5: return v3(vo_max(cb.x, cs.x), vo_max(cb.y, cs.y), vo_max(cb.z, cs.z));
    case 6: return v3(color_dodge(cb.x, cs.x), color_dodge(cb.y, cs.y), color_dodge(cb.z, cs.z));
    case 7: return v3(color_burn(cb.x, cs.x), color_burn(cb.y, cs.y), color_burn(cb.z, cs.z));
    case 8: return v3(hard_light1(cb.x, cs.x), hard_light1(cb.y, cs.y), hard_light1(cb.z, cs.z));
    case 9: return v3(soft_light1(cb.x, cs.x), soft_light1(cb.y, cs.y), soft_light1(cb.z, cs.z));
    case 10: return v3(fabsf(cb.x - cs.x), fabsf(cb.y - cs.y), fabsf(cb.z - cs.z));
    case 11: return v3(cb.x + cs.x - 2.0f * cb.x * cs.x, cb.y + cs.y - 2.0f * cb.y * cs.y, cb.z + cs.z - 2.0f * cb.z * cs.z);
    case 12: return set_lum(set_sat(cs, sat(cb)), lum(cb));
    case 13: return set_lum(set_sat(cb, sat(cs)), lum(cb));
    case 14: return set_lum(cs, lum(cb));
    case 15: return set_lum(cb, lum(cs));
    default: return cs;
    }
}
static vec4 blend_compose(vec3 cb, vec3 cs, float ab, float as_, uint32_t mode) {
    float fa = 0.0f, fb = 0.0f;
    switch (mode) {
    case 1: fa = 1.0f; fb = 0.0f; break;
    case 2: fa = 0.0f; fb = 1.0f; break;
    case 3: fa = 1.0f; fb = 1.0f - as_; break;
    case 4: fa = 1.0f - ab; fb = 1.0f; break;
    case 5: fa = ab; fb = 0.0f; break;
    case 6: fa = 0.0f; fb = as_; break;
    case 7: fa = 1.0f - ab; fb = 0.0f; break;
    case 8: fa = 0.0f; fb = 1.0f - as_; break;
    case 9: fa = ab; fb = 1.0f - as_; break;
    case 10: fa = 1.0f - ab; fb = as_; break;
    case 11: fa = 1.0f - ab; fb = 1.0f - as_; break;
    case 12: fa = 1.0f; fb = 1.0f; break;
    case 13: {
        vec4 r;
        r.v[0] = vo_min(1.0f, as_ * cs.x + ab * cb.x);
        r.v[1] = vo_min(1.0f, as_ * cs.y + ab * cb.y);
        r.v[2] = vo_min(1.0f, as_ * cs.z + ab * cb.z);
        r.v[3] = vo_min(1.0f, as_ + ab);
        return r;
    }
    default: break;
    }
    float as_fa = as_ * fa, ab_fb = ab * fb;
    vec4 r;
    r.v[0] = as_fa * cs.x + ab_fb * cb.x;
    r.v[1] = as_fa * cs.y + ab_fb * cb.y;
    r.v[2] = as_fa * cs.z + ab_fb * cb.z;
    r.v[3] = vo_min(as_fa + ab_fb, 1.0f);
    return r;
}
static vec3 unpremultiply(vec4 c) {
    float inv_alpha = 1.0f / vo_max(c.v[3], 1e-15f);
    return v3(c.v[0] * inv_alpha, c.v[1] * inv_alpha, c.v[2] * inv_alpha);
}
static float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static vec4 blend_mix_compose(vec4 backdrop, vec4 src, uint32_t mode) {
    const uint32_t BLEND_DEFAULT = (0u << 8) | 3u;
    vec4 r;
    if ((mode & 0x7fffu) == BLEND_DEFAULT) {
        for (int k = 0; k < 4; k++) r.v[k] = backdrop.v[k] * (1.0f - src.v[3]) + src.v[k];
        return r;
    }
    vec3 cs = unpremultiply(src);
    vec3 cb = unpremultiply(backdrop);
    uint32_t mix_mode = mode >> 8;
    vec3 mixed = blend_mix(cb, cs, mix_mode);
    cs = v3(mixf(cs.x, mixed.x, backdrop.v[3]), mixf(cs.y, mixed.y, backdrop.v[3]), mixf(cs.z, mixed.z, backdrop.v[3]));
    uint32_t compose_mode = mode & 0xffu;
    if (compose_mode == 3u) {
        r.v[0] = mixf(backdrop.v[0], cs.x, src.v[3]);
        r.v[1] = mixf(backdrop.v[1], cs.y, src.v[3]);
        r.v[2] = mixf(backdrop.v[2], cs.z, src.v[3]);
        r.v[3] = src.v[3] + backdrop.v[3] * (1.0f - src.v[3]);
        return r;
    }
    return blend_compose(cb, cs, backdrop.v[3], src.v[3], compose_mode);
}

/* fine.wgsl:863-875 */
static float extend_mode_normalized(float t, uint32_t mode) {
    switch (mode) {
    case 0: return vo_clamp(t, 0.0f, 1.0f);
    case 1: return t - floorf(t);
    default: return fabsf(t - 2.0f * vo_round(0.5f * t));
    }
}
static vec4 ramp_load(const vo_ctx *c, int32_t x, uint32_t index) {
    vec4 z = {{0, 0, 0, 0}};
    if (!c->ramps || index >= c->n_ramps || x < 0 || x >= GRADIENT_WIDTH) return z;
    return unpack4x8unorm(c->ramps[index * GRADIENT_WIDTH + (uint32_t)x]);
}
static void src_over(vec4 *rgba, vec4 fg, float area) {
    float fa = fg.v[3] * area;
    for (int k = 0; k < 4; k++) rgba->v[k] = rgba->v[k] * (1.0f - fa) + fg.v[k] * area;
}

/* ---------------- images: fine.wgsl:804-993 ---------------- */
/* textureLoad(image_atlas, vec2<i32>(uv), 0): out-of-range loads return transparent black */
static vec4 atlas_load(const vo_ctx *c, float u, float v, uint32_t alpha_type) {
    vec4 z = {{0, 0, 0, 0}};
    int32_t x = f2i(truncf(u)), y = f2i(truncf(v));
    if (!c->atlas || x < 0 || y < 0 || (uint32_t)x >= c->atlas_w || (uint32_t)y >= c->atlas_h) return z;
    vec4 p = unpack4x8unorm(c->atlas[(size_t)y * c->atlas_w + (uint32_t)x]);
    if (alpha_type == 0u) { /* maybe_premul_alpha, fine.wgsl:847-858 */
        p.v[0] *= p.v[3]; p.v[1] *= p.v[3]; p.v[2] *= p.v[3];
    }
    return p;
}
static float extend_mode_px(float t, uint32_t mode, float max) { /* fine.wgsl:877-889 */
    if (mode == 0u) return vo_clamp(t, 0.0f, max);
    return extend_mode_normalized(t / max, mode) * max;
}
static float single_weight(float t, float a, float b, float c, float d) { return t * (t * (t * d + c) + b) + a; }
static void cubic_weights(float fr, float w[4]) { /* fine.wgsl:893-931, Mitchell B = C = 1/3 */
    static const float MF[4][4] = {
        {(1.0f / 6.0f) / 3.0f, -(3.0f / 6.0f) / 3.0f - 1.0f / 3.0f, (3.0f / 6.0f) / 3.0f + 2.0f * 1.0f / 3.0f, -(1.0f / 6.0f) / 3.0f - 1.0f / 3.0f},
        {1.0f - (2.0f / 6.0f) / 3.0f, 0.0f, -3.0f + (12.0f / 6.0f) / 3.0f + 1.0f / 3.0f, 2.0f - (9.0f / 6.0f) / 3.0f - 1.0f / 3.0f},
        {(1.0f / 6.0f) / 3.0f, (3.0f / 6.0f) / 3.0f + 1.0f / 3.0f, 3.0f - (15.0f / 6.0f) / 3.0f - 2.0f * 1.0f / 3.0f, -2.0f + (9.0f / 6.0f) / 3.0f + 1.0f / 3.0f},
        {0.0f, 0.0f, -1.0f / 3.0f, (1.0f / 6.0f) / 3.0f + 1.0f / 3.0f}};
    for (int k = 0; k < 4; k++) w[k] = single_weight(fr, MF[k][0], MF[k][1], MF[k][2], MF[k][3]);
}
static vec4 bicubic_sample(const vo_ctx *c, float cx_, float cy_, float ox, float oy, float mx, float my, uint32_t alpha_type) {
    float fx = (cx_ + 0.5f) - floorf(cx_ + 0.5f), fy = (cy_ + 0.5f) - floorf(cy_ + 0.5f);
    float wx[4], wy[4];
    cubic_weights(fx, wx);
    cubic_weights(fy, wy);
    static const float off[4] = {-1.5f, -0.5f, 0.5f, 1.5f};
    vec4 res = {{0, 0, 0, 0}};
    vec4 rows[4];
    for (int j = 0; j < 4; j++) {
        vec4 s[4];
        for (int i = 0; i < 4; i++)
            s[i] = atlas_load(c, vo_clamp(cx_ + off[i], ox, mx), vo_clamp(cy_ + off[j], oy, my), alpha_type);
        for (int k = 0; k < 4; k++) rows[j].v[k] = wx[0] * s[0].v[k] + wx[1] * s[1].v[k] + wx[2] * s[2].v[k] + wx[3] * s[3].v[k];
    }
    for (int k = 0; k < 4; k++) res.v[k] = wy[0] * rows[0].v[k] + wy[1] * rows[1].v[k] + wy[2] * rows[2].v[k] + wy[3] * rows[3].v[k];
    float a = vo_clamp(res.v[3], 0.0f, 1.0f);
    for (int k = 0; k < 3; k++) res.v[k] = vo_clamp(res.v[k], 0.0f, a);
    res.v[3] = a;
    return res;
}
/* fine.wgsl:715-726 */
static float erf7(float x) {
    float y = vo_clamp(x * 1.1283791671f, -100.0f, 100.0f);
    float yy = y * y;
    float z = y + (0.24295f + (0.03395f + 0.0104f * yy) * yy) * (y * yy);
    return z / sqrtf(1.0f + z * z);
}
static float hypot_wgsl(float a, float b) { return sqrtf(a * a + b * b); }

/* ---------------- tile interpreter: fine.wgsl:1064-1398 ---------------- */
static void fine_tile(const vo_ctx *c, fine_shared *sh, uint32_t tile_x, uint32_t tile_y) {
    const vo_config *cfg = &c->cfg;
    const uint32_t *ptcl = (const uint32_t *)c->buf[VO_BUF_PTCL];
    const uint32_t *info = (const uint32_t *)c->buf[VO_BUF_INFO_BIN_DATA];
    uint32_t *blend_spill = (uint32_t *)c->buf[VO_BUF_BLEND_SPILL];
    uint8_t *output = (uint8_t *)c->buf[VO_BUF_OUTPUT];
    uint32_t tile_ix = tile_y * cfg->width_in_tiles + tile_x;
    vec4 rgba[256];
    uint32_t blend_stack[BLEND_STACK_SPLIT][256];
    float area[256];
    vec4 base_color = unpack4x8unorm(cfg->base_color);
    for (uint32_t i = 0; i < 256; i++) { rgba[i] = base_color; area[i] = 0.0f; }
    uint32_t clip_depth = 0u;
    uint32_t cmd_ix = tile_ix * PTCL_INITIAL_ALLOC;
    uint32_t blend_offset = ptcl[cmd_ix];
    cmd_ix += 1u;
    for (;;) {
        uint32_t tag = ptcl[cmd_ix];
        if (tag == CMD_END) break;
        switch (tag) {
        case CMD_FILL: {
            cmd_fill fill;
            fill.size_and_rule = ptcl[cmd_ix + 1u];
            fill.seg_data = ptcl[cmd_ix + 2u];
            fill.backdrop = (int32_t)ptcl[cmd_ix + 3u];
            if (c->aa == VO_AA_AREA) fill_path_area(c, fill, area);
            else fill_path_ms(c, sh, fill, c->aa == VO_AA_MSAA16, area);
            cmd_ix += 4u;
            break;
        }
        case CMD_SOLID:
            for (uint32_t i = 0; i < 256; i++) area[i] = 1.0f;
            cmd_ix += 1u;
            break;
        case CMD_COLOR: {
            vec4 fg = unpack4x8unorm(ptcl[cmd_ix + 1u]);
            for (uint32_t i = 0; i < 256; i++) {
                vec4 fg_i;
                for (int k = 0; k < 4; k++) fg_i.v[k] = fg.v[k] * area[i];
                for (int k = 0; k < 4; k++) rgba[i].v[k] = rgba[i].v[k] * (1.0f - fg_i.v[3]) + fg_i.v[k];
            }
            cmd_ix += 2u;
            break;
        }
        case CMD_BEGIN_CLIP:
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t packed = pack4x8unorm(rgba[i]);
                if (clip_depth < BLEND_STACK_SPLIT) {
                    blend_stack[clip_depth][i] = packed;
                } else {
                    uint32_t blend_in_scratch = clip_depth - BLEND_STACK_SPLIT;
                    uint32_t ix = blend_offset + blend_in_scratch * TILE_WIDTH * TILE_HEIGHT + i;
                    if (ix < cfg->blend_size) blend_spill[ix] = packed;
                }
                for (int k = 0; k < 4; k++) rgba[i].v[k] = 0.0f;
            }
            clip_depth += 1u;
            cmd_ix += 1u;
            break;
        case CMD_END_CLIP: {
            uint32_t blend = ptcl[cmd_ix + 1u];
            float alpha = bits2f(ptcl[cmd_ix + 2u]);
            clip_depth -= 1u;
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t bg_rgba;
                if (clip_depth < BLEND_STACK_SPLIT) {
                    bg_rgba = blend_stack[clip_depth][i];
                } else {
                    uint32_t blend_in_scratch = clip_depth - BLEND_STACK_SPLIT;
                    uint32_t ix = blend_offset + blend_in_scratch * TILE_WIDTH * TILE_HEIGHT + i;
                    bg_rgba = ix < cfg->blend_size ? blend_spill[ix] : 0u;
                }
                vec4 bg = unpack4x8unorm(bg_rgba);
                vec4 fg;
                for (int k = 0; k < 4; k++) fg.v[k] = rgba[i].v[k] * area[i] * alpha;
                if (blend == LUMINANCE_MASK_LAYER) {
                    if (area[i] == 0.0f) { rgba[i] = bg; continue; }
                    float luminance = vo_clamp(svg_lum(unpremultiply(fg)) * fg.v[3], 0.0f, 1.0f);
                    for (int k = 0; k < 4; k++) rgba[i].v[k] = bg.v[k] * luminance;
                } else {
                    rgba[i] = blend_mix_compose(bg, fg, blend);
                }
            }
            cmd_ix += 3u;
            break;
        }
        case CMD_JUMP:
            cmd_ix = ptcl[cmd_ix + 1u];
            break;
        case CMD_LIN_GRAD: {
            uint32_t index_mode = ptcl[cmd_ix + 1u];
            uint32_t index = index_mode >> 2, extend = index_mode & 3u;
            uint32_t io = ptcl[cmd_ix + 2u];
            float line_x = bits2f(info[io]), line_y = bits2f(info[io + 1u]), line_c = bits2f(info[io + 2u]);
            for (uint32_t th = 0; th < 64; th++) {
                float xy_x = (float)(tile_x * TILE_WIDTH + (th & 3u) * 4u), xy_y = (float)(tile_y * TILE_HEIGHT + (th >> 2));
                float d = line_x * xy_x + line_y * xy_y + line_c;
                for (uint32_t k = 0; k < 4; k++) {
                    float my_d = d + line_x * (float)k;
                    int32_t x = f2i(vo_round(extend_mode_normalized(my_d, extend) * (float)(GRADIENT_WIDTH - 1)));
                    src_over(&rgba[th * 4u + k], ramp_load(c, x, index), area[th * 4u + k]);
                }
            }
            cmd_ix += 3u;
            break;
        }
        case CMD_RAD_GRAD: {
            uint32_t index_mode = ptcl[cmd_ix + 1u];
            uint32_t index = index_mode >> 2, extend = index_mode & 3u;
            uint32_t io = ptcl[cmd_ix + 2u];
            float m0 = bits2f(info[io]), m1 = bits2f(info[io + 1u]), m2 = bits2f(info[io + 2u]), m3 = bits2f(info[io + 3u]);
            float xl0 = bits2f(info[io + 4u]), xl1 = bits2f(info[io + 5u]);
            float focal_x = bits2f(info[io + 6u]), radius = bits2f(info[io + 7u]);
            uint32_t flags_kind = info[io + 8u];
            uint32_t flags = flags_kind >> 3, kind = flags_kind & 7u;
            int is_strip = kind == RAD_GRAD_KIND_STRIP, is_circular = kind == RAD_GRAD_KIND_CIRCULAR;
            int is_focal_on_circle = kind == RAD_GRAD_KIND_FOCAL_ON_CIRCLE;
            int is_swapped = (flags & RAD_GRAD_SWAPPED) != 0u;
            float r1_recip = is_circular ? 0.0f : 1.0f / radius;
            float less_scale = (is_swapped || (1.0f - focal_x) < 0.0f) ? -1.0f : 1.0f;
            float t_sign = vo_sign(1.0f - focal_x);
            for (uint32_t i = 0; i < 256; i++) {
                float mx = (float)(tile_x * TILE_WIDTH + ((i >> 2) & 3u) * 4u) + (float)(i & 3u);
                float my = (float)(tile_y * TILE_HEIGHT + (i >> 4));
                float x = m0 * mx + m2 * my + xl0;
                float y = m1 * mx + m3 * my + xl1;
                float xx = x * x, yy = y * y;
                float t = 0.0f;
                int is_valid = 1;
                if (is_strip) {
                    float a = radius - yy;
                    t = sqrtf(a) + x;
                    is_valid = a >= 0.0f;
                } else if (is_focal_on_circle) {
                    t = (xx + yy) / x;
                    is_valid = t >= 0.0f && x != 0.0f;
                } else if (radius > 1.0f) {
                    t = sqrtf(xx + yy) - x * r1_recip;
                } else {
                    float a = xx - yy;
                    t = less_scale * sqrtf(a) - x * r1_recip;
                    is_valid = a >= 0.0f && t >= 0.0f;
                }
                if (is_valid) {
                    t = extend_mode_normalized(focal_x + t_sign * t, extend);
                    if (is_swapped) t = 1.0f - t;
                    int32_t rx = f2i(vo_round(t * (float)(GRADIENT_WIDTH - 1)));
                    src_over(&rgba[i], ramp_load(c, rx, index), area[i]);
                }
            }
            cmd_ix += 3u;
            break;
        }
        case CMD_SWEEP_GRAD: {
            uint32_t index_mode = ptcl[cmd_ix + 1u];
            uint32_t index = index_mode >> 2, extend = index_mode & 3u;
            uint32_t io = ptcl[cmd_ix + 2u];
            float m0 = bits2f(info[io]), m1 = bits2f(info[io + 1u]), m2 = bits2f(info[io + 2u]), m3 = bits2f(info[io + 3u]);
            float xl0 = bits2f(info[io + 4u]), xl1 = bits2f(info[io + 5u]);
            float t0 = bits2f(info[io + 6u]), t1 = bits2f(info[io + 7u]);
            float scale = 1.0f / (t1 - t0);
            for (uint32_t i = 0; i < 256; i++) {
                float mx = (float)(tile_x * TILE_WIDTH + ((i >> 2) & 3u) * 4u) + (float)(i & 3u);
                float my = (float)(tile_y * TILE_HEIGHT + (i >> 4));
                float x = m0 * mx + m2 * my + xl0;
                float y = m1 * mx + m3 * my + xl1;
                float xabs = fabsf(x), yabs = fabsf(y);
                float slope = vo_min(xabs, yabs) / vo_max(xabs, yabs);
                float s = slope * slope;
                float phi = slope * (0.15912117063999176025390625f + s * (-5.185396969318389892578125e-2f + s * (2.476101927459239959716796875e-2f + s * (-7.0547382347285747528076171875e-3f))));
                if (xabs < yabs) phi = 1.0f / 4.0f - phi;
                if (x < 0.0f) phi = 1.0f / 2.0f - phi;
                if (y < 0.0f) phi = 1.0f - phi;
                if (phi != phi) phi = 0.0f;
                phi = (phi - t0) * scale;
                float t = extend_mode_normalized(phi, extend);
                int32_t rx = f2i(vo_round(t * (float)(GRADIENT_WIDTH - 1)));
                src_over(&rgba[i], ramp_load(c, rx, index), area[i]);
            }
            cmd_ix += 3u;
            break;
        }
        case CMD_IMAGE: { /* fine.wgsl:804-827 (read_image), :1315-1382 */
            uint32_t io = ptcl[cmd_ix + 1u];
            float m0 = bits2f(info[io]), m1 = bits2f(info[io + 1u]), m2 = bits2f(info[io + 2u]), m3 = bits2f(info[io + 3u]);
            float xl0 = bits2f(info[io + 4u]), xl1 = bits2f(info[io + 5u]);
            uint32_t xy = info[io + 6u], width_height = info[io + 7u], sample_alpha = info[io + 8u];
            float alpha = (float)(sample_alpha & 0xFFu) / 255.0f;
            uint32_t format = sample_alpha >> 15, alpha_type = (sample_alpha >> 14) & 1u, quality = (sample_alpha >> 12) & 3u;
            uint32_t x_extend = (sample_alpha >> 10) & 3u, y_extend = (sample_alpha >> 8) & 3u;
            float ox = (float)(xy >> 16), oy = (float)(xy & 0xffffu);
            float ew = (float)(width_height >> 16), eh = (float)(width_height & 0xffffu);
            float amx = ox + ew - 1.0f, amy = oy + eh - 1.0f;
            for (uint32_t i = 0; i < 256; i++) {
                if (area[i] == 0.0f) continue;
                float px = (float)(tile_x * TILE_WIDTH + (i & 15u)) + 0.5f;
                float py = (float)(tile_y * TILE_HEIGHT + (i >> 4)) + 0.5f;
                float u = m0 * px + m2 * py + xl0;
                float v = m1 * px + m3 * py + xl1;
                u = extend_mode_px(u, x_extend, ew);
                v = extend_mode_px(v, y_extend, eh);
                vec4 fg;
                if (quality == 0u) {
                    u += ox; v += oy;
                    fg = atlas_load(c, vo_clamp(u, ox, amx), vo_clamp(v, oy, amy), alpha_type);
                } else if (quality == 2u) {
                    u += ox; v += oy;
                    fg = bicubic_sample(c, u, v, ox, oy, amx, amy, alpha_type);
                } else {
                    u = u + ox - 0.5f; v = v + oy - 0.5f;
                    float uc = vo_clamp(u, ox, amx), vc = vo_clamp(v, oy, amy);
                    float x0 = floorf(uc), y0 = floorf(vc), x1 = ceilf(uc), y1 = ceilf(vc);
                    float frx = u - floorf(u), fry = v - floorf(v);
                    vec4 a = atlas_load(c, x0, y0, alpha_type), b = atlas_load(c, x0, y1, alpha_type);
                    vec4 cc = atlas_load(c, x1, y0, alpha_type), d = atlas_load(c, x1, y1, alpha_type);
                    for (int k = 0; k < 4; k++) fg.v[k] = mixf(mixf(a.v[k], b.v[k], fry), mixf(cc.v[k], d.v[k], fry), frx);
                }
                vec4 fg_i;
                for (int k = 0; k < 4; k++) fg_i.v[k] = fg.v[k] * area[i] * alpha;
                if (format == 1u) { float t = fg_i.v[0]; fg_i.v[0] = fg_i.v[2]; fg_i.v[2] = t; } /* pixel_format: .bgra */
                for (int k = 0; k < 4; k++) rgba[i].v[k] = rgba[i].v[k] * (1.0f - fg_i.v[3]) + fg_i.v[k];
            }
            cmd_ix += 2u;
            break;
        }
        case CMD_BLUR_RECT: { /* fine.wgsl:740-756 (read_blur_rect), :1173-1224 */
            uint32_t io = ptcl[cmd_ix + 1u];
            vec4 blur_rgba = unpack4x8unorm(ptcl[cmd_ix + 2u]);
            float m0 = bits2f(info[io]), m1 = bits2f(info[io + 1u]), m2 = bits2f(info[io + 2u]), m3 = bits2f(info[io + 3u]);
            float xl0 = bits2f(info[io + 4u]), xl1 = bits2f(info[io + 5u]);
            float bw = bits2f(info[io + 6u]), bh = bits2f(info[io + 7u]), bradius = bits2f(info[io + 8u]), bstd = bits2f(info[io + 9u]);
            float std_dev = vo_max(bstd, 1e-5f);
            float inv_std_dev = 1.0f / std_dev;
            float min_edge = vo_min(bw, bh);
            float radius_max = 0.5f * min_edge;
            float r0 = vo_min(hypot_wgsl(bradius, std_dev * 1.15f), radius_max);
            float r1 = vo_min(hypot_wgsl(bradius, std_dev * 2.0f), radius_max);
            float exponent = 2.0f * r1 / r0;
            float inv_exponent = 1.0f / exponent;
            float delta = 1.25f * std_dev * (vo_expf(-vo_powf(0.5f * inv_std_dev * bw, 2.0f)) - vo_expf(-vo_powf(0.5f * inv_std_dev * bh, 2.0f)));
            float width = bw + vo_min(delta, 0.0f);
            float height = bh - vo_max(delta, 0.0f);
            float scale = 0.5f * erf7(inv_std_dev * 0.5f * (vo_max(width, height) - 0.5f * bradius));
            for (uint32_t i = 0; i < 256; i++) {
                float px = (float)(tile_x * TILE_WIDTH + (i & 15u));
                float py = (float)(tile_y * TILE_HEIGHT + (i >> 4));
                float x = m0 * px + m2 * py + xl0;
                float y = m1 * px + m3 * py + xl1;
                float y0 = fabsf(y) - (height * 0.5f - r1);
                float y1 = vo_max(y0, 0.0f);
                float x0 = fabsf(x) - (width * 0.5f - r1);
                float x1 = vo_max(x0, 0.0f);
                float d_pos = vo_powf(vo_powf(x1, exponent) + vo_powf(y1, exponent), inv_exponent);
                float d_neg = vo_min(vo_max(x0, y0), 0.0f);
                float d = d_pos + d_neg - r1;
                float alpha = scale * (erf7(inv_std_dev * (min_edge + d)) - erf7(inv_std_dev * d));
                vec4 fg;
                for (int k = 0; k < 4; k++) fg.v[k] = blur_rgba.v[k] * alpha;
                src_over(&rgba[i], fg, area[i]);
            }
            cmd_ix += 3u;
            break;
        }
        default: cmd_ix += 1u; break;
        }
    }
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t px = tile_x * TILE_WIDTH + (i & 15u), py = tile_y * TILE_HEIGHT + (i >> 4);
        if (px < cfg->target_width && py < cfg->target_height) {
            vec4 fg = rgba[i];
            float a_inv = 1.0f / vo_max(fg.v[3], 1e-6f);
            vec4 sep;
            sep.v[0] = fg.v[0] * a_inv; sep.v[1] = fg.v[1] * a_inv; sep.v[2] = fg.v[2] * a_inv; sep.v[3] = fg.v[3];
            uint32_t packed = pack4x8unorm(sep);
            memcpy(output + ((size_t)py * cfg->target_width + px) * 4u, &packed, 4);
        }
    }
}

typedef struct { vo_ctx *c; uint32_t k, n; } fine_job;

static void *fine_worker(void *arg) {
    fine_job *j = (fine_job *)arg;
    const vo_config *cfg = &j->c->cfg;
    fine_shared sh;
    uint32_t n_tiles = cfg->width_in_tiles * cfg->height_in_tiles;
    for (uint32_t t = j->k; t < n_tiles; t += j->n)
        fine_tile(j->c, &sh, t % cfg->width_in_tiles, t / cfg->width_in_tiles);
    return NULL;
}

void vo_stage_fine(vo_ctx *c) {
    const uint32_t *ptcl = (const uint32_t *)c->buf[VO_BUF_PTCL];
    if (ptcl[0] == ~0u) return; /* fine.wgsl:1070-1074 */
    uint32_t n = c->n_threads > 1 ? (uint32_t)c->n_threads : 1u;
    if (n == 1u) {
        fine_job j = {c, 0, 1};
        fine_worker(&j);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n);
    fine_job *jobs = (fine_job *)malloc(sizeof(fine_job) * n);
    for (uint32_t k = 0; k < n; k++) {
        jobs[k].c = c; jobs[k].k = k; jobs[k].n = n;
        pthread_create(&th[k], NULL, fine_worker, &jobs[k]);
    }
    for (uint32_t k = 0; k < n; k++) pthread_join(th[k], NULL);
    free(th);
    free(jobs);
}
