/*
 * Oracle front half: pathtag scan, bbox_clear, flatten, draw scan, clip.
 * TEST INFRASTRUCTURE ONLY (see vello_oracle.h).
 */
#include "vo_internal.h"

/* ------------------------------------------------------------------ */
/* pathtag_reduce + pathtag_scan + bbox_clear                          */
/* vello_shaders/src/cpu/pathtag_reduce.rs:12-27, pathtag_scan.rs:12-32,
 * bbox_clear.rs:8-15; monoid: vello_encoding/src/path.rs:338-364,
 * shader/shared/pathtag.wgsl:58-71                                    */
/* ------------------------------------------------------------------ */
vo_tag_monoid vo_reduce_tag(uint32_t tag_word) {
    vo_tag_monoid c;
    uint32_t point_count = tag_word & 0x3030303u;
    c.pathseg_ix = popcnt((point_count * 7u) & 0x4040404u);
    c.trans_ix = popcnt(tag_word & (PATH_TAG_TRANSFORM * 0x1010101u));
    uint32_t n_points = point_count + ((tag_word >> 2) & 0x1010101u);
    uint32_t a = n_points + (n_points & (((tag_word >> 3) & 0x1010101u) * 15u));
    a += a >> 8;
    a += a >> 16;
    c.pathseg_offset = a & 0xffu;
    c.path_ix = popcnt(tag_word & (PATH_TAG_PATH * 0x1010101u));
    c.style_ix = popcnt(tag_word & (PATH_TAG_STYLE * 0x1010101u)) * STYLE_SIZE_IN_WORDS;
    return c;
}

static vo_tag_monoid combine_tag(vo_tag_monoid a, vo_tag_monoid b) {
    vo_tag_monoid c;
    c.trans_ix = a.trans_ix + b.trans_ix;
    c.pathseg_ix = a.pathseg_ix + b.pathseg_ix;
    c.pathseg_offset = a.pathseg_offset + b.pathseg_offset;
    c.style_ix = a.style_ix + b.style_ix;
    c.path_ix = a.path_ix + b.path_ix;
    return c;
}

void vo_stage_pathtag_scan(vo_ctx *c) {
    /* render.rs:313: bump buffer is cleared before flatten */
    memset(c->buf[VO_BUF_BUMP], 0, sizeof(vo_bump));
    vo_tag_monoid *tm = (vo_tag_monoid *)c->buf[VO_BUF_TAG_MONOIDS];
    const uint32_t *tags = c->scene + c->cfg.layout.path_tag_base;
    vo_tag_monoid m;
    memset(&m, 0, sizeof m);
    for (uint32_t i = 0; i < c->n_tag_words; i++) {
        tm[i] = m; /* exclusive prefix per 4-tag word */
        m = combine_tag(m, vo_reduce_tag(tags[i]));
    }
    vo_path_bbox *bb = (vo_path_bbox *)c->buf[VO_BUF_PATH_BBOXES];
    for (uint32_t i = 0; i < c->cfg.layout.n_paths; i++) {
        bb[i].x0 = 0x7fffffff;
        bb[i].y0 = 0x7fffffff;
        bb[i].x1 = (int32_t)0x80000000;
        bb[i].y1 = (int32_t)0x80000000;
    }
}

/* ------------------------------------------------------------------ */
/* flatten: vello_shaders/shader/flatten.wgsl (source of truth),       */
/* cpu/flatten.rs + cpu/euler.rs for structure                         */
/* ------------------------------------------------------------------ */
#define DERIV_THRESH 1e-6f
#define DERIV_THRESH_SQUARED (DERIV_THRESH * DERIV_THRESH)
#define DERIV_EPS 1e-6f
#define SUBDIV_LIMIT (1.0f / 65536.0f)
#define K1_THRESH 1e-3f
#define DIST_THRESH 1e-3f
#define TANGENT_THRESH 1e-6f

typedef struct { float th0, th1, chord_len, err; } cubic_params;
typedef struct { float th0, k0, k1, ch; } euler_params;
typedef struct { vec2 p0, p1, p2, p3; } cubic_points;
typedef struct { vec2 point, deriv; } point_deriv;

typedef struct {
    vo_ctx *c;
    vo_line_soup *lines;
    vo_bump *bump;
    float bbox[4]; /* per-invocation bbox (flatten.wgsl:828) */
} flat_state;

/* flatten.wgsl:668-672 (uses fma) */
static vec2 xf_apply(const vo_xform *t, vec2 p) {
    float px = fmaf(t->m[0], p.x, fmaf(t->m[2], p.y, t->t[0]));
    float py = fmaf(t->m[1], p.x, fmaf(t->m[3], p.y, t->t[1]));
    return v2(px, py);
}

/* flatten.wgsl:766-773 */
static void write_line(flat_state *s, uint32_t line_ix, uint32_t path_ix, vec2 p0, vec2 p1) {
    s->bbox[0] = vo_min(s->bbox[0], vo_min(p0.x, p1.x));
    s->bbox[1] = vo_min(s->bbox[1], vo_min(p0.y, p1.y));
    s->bbox[2] = vo_max(s->bbox[2], vo_max(p0.x, p1.x));
    s->bbox[3] = vo_max(s->bbox[3], vo_max(p0.y, p1.y));
    if (line_ix < s->c->cfg.lines_size) {
        vo_line_soup *l = &s->lines[line_ix];
        l->path_ix = path_ix;
        l->pad = 0;
        l->p0[0] = p0.x; l->p0[1] = p0.y;
        l->p1[0] = p1.x; l->p1[1] = p1.y;
    }
}
static void write_line_xf(flat_state *s, uint32_t line_ix, uint32_t path_ix, vec2 p0, vec2 p1, const vo_xform *t) {
    write_line(s, line_ix, path_ix, xf_apply(t, p0), xf_apply(t, p1));
}
static uint32_t alloc_lines(flat_state *s, uint32_t n) {
    /* atomic for the threaded CPU-baseline mode (vo_set_threads); with one thread it is the plain bump of the shader */
    return __atomic_fetch_add(&s->bump->lines, n, __ATOMIC_RELAXED);
}
static void atomic_min_i32(int32_t *p, int32_t v) {
    int32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
static void atomic_max_i32(int32_t *p, int32_t v) {
    int32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
static void output_line_xf(flat_state *s, uint32_t path_ix, vec2 p0, vec2 p1, const vo_xform *t) {
    write_line_xf(s, alloc_lines(s, 1), path_ix, p0, p1, t);
}
static void output_two_lines_xf(flat_state *s, uint32_t path_ix, vec2 p00, vec2 p01, vec2 p10, vec2 p11, const vo_xform *t) {
    uint32_t ix = alloc_lines(s, 2);
    write_line_xf(s, ix, path_ix, p00, p01, t);
    write_line_xf(s, ix + 1, path_ix, p10, p11, t);
}

/* flatten.wgsl:94-133 */
static cubic_params cubic_from_points_derivs(vec2 p0, vec2 p1, vec2 q0, vec2 q1, float dt) {
    cubic_params r;
    vec2 chord = vsub(p1, p0);
    float chord_squared = vdot(chord, chord);
    float chord_len = sqrtf(chord_squared);
    if (chord_squared < DERIV_THRESH_SQUARED) {
        float chord_err = sqrtf((9.0f / 32.0f) * (vdot(q0, q0) + vdot(q1, q1))) * dt;
        r.th0 = 0.0f; r.th1 = 0.0f; r.chord_len = DERIV_THRESH; r.err = chord_err;
        return r;
    }
    float scale = dt / chord_squared;
    vec2 h0 = v2(q0.x * chord.x + q0.y * chord.y, q0.y * chord.x - q0.x * chord.y);
    float th0 = vo_atan2f(h0.y, h0.x);
    float d0 = vlen(h0) * scale;
    vec2 h1 = v2(q1.x * chord.x + q1.y * chord.y, q1.x * chord.y - q1.y * chord.x);
    float th1 = vo_atan2f(h1.y, h1.x);
    float d1 = vlen(h1) * scale;
    float cth0 = vo_cosf(th0);
    float cth1 = vo_cosf(th1);
    float err = 2.0f;
    if (cth0 * cth1 >= 0.0f) {
        float e0 = (2.0f / 3.0f) / vo_max(1.0f + cth0, 1e-9f);
        float e1 = (2.0f / 3.0f) / vo_max(1.0f + cth1, 1e-9f);
        float s0 = vo_sinf(th0);
        float s1 = vo_sinf(th1);
        float s01 = cth0 * s1 + cth1 * s0;
        float amin = 0.15f * (2.0f * e0 * s0 + 2.0f * e1 * s1 - e0 * e1 * s01);
        float a = 0.15f * (2.0f * d0 * s0 + 2.0f * d1 * s1 - d0 * d1 * s01);
        float aerr = fabsf(a - amin);
        float symm = fabsf(th0 + th1);
        float asymm = fabsf(th0 - th1);
        float dist = vlen(v2(d0 - e0, d1 - e1));
        float symm2 = symm * symm;
        float ctr = (4.625e-6f * symm * symm2 + 7.5e-3f * asymm) * symm2;
        float halo = (5e-3f * symm + 7e-2f * asymm) * dist;
        err = ctr + 1.55f * aerr + halo;
    }
    err *= chord_len;
    r.th0 = th0; r.th1 = th1; r.chord_len = chord_len; r.err = err;
    return r;
}

/* flatten.wgsl:135-158 */
static euler_params es_params_from_angles(float th0, float th1) {
    euler_params r;
    float k0 = th0 + th1;
    float dth = th1 - th0;
    float d2 = dth * dth;
    float k2 = k0 * k0;
    float a = 6.0f;
    a -= d2 * (1.0f / 70.0f);
    a -= (d2 * d2) * (1.0f / 10780.0f);
    a += (d2 * d2 * d2) * 2.769178184818219e-07f;
    float b = -0.1f + d2 * (1.0f / 4200.0f) + d2 * d2 * 1.6959677820260655e-05f;
    float cc = -1.0f / 1400.0f + d2 * 6.84915970574303e-05f - k2 * 7.936475029053326e-06f;
    a += (b + cc * k2) * k2;
    float k1 = dth * a;
    float ch = 1.0f;
    ch -= d2 * (1.0f / 40.0f);
    ch += (d2 * d2) * 0.00034226190482569864f;
    ch -= (d2 * d2 * d2) * 1.9349474568904524e-06f;
    float b_ = -1.0f / 24.0f + d2 * 0.0024702380951963226f - d2 * d2 * 3.7297408997537985e-05f;
    float c_ = 1.0f / 1920.0f - d2 * 4.87350869747975e-05f - k2 * 3.1001936068463107e-06f;
    ch += (b_ + c_ * k2) * k2;
    r.th0 = th0; r.k0 = k0; r.k1 = k1; r.ch = ch;
    return r;
}

static float es_params_eval_th(const euler_params *p, float t) {
    return (p->k0 + 0.5f * p->k1 * (t - 1.0f)) * t - p->th0;
}

/* flatten.wgsl:165-196 */
static vec2 integ_euler_10(float k0, float k1) {
    float t1_1 = k0;
    float t1_2 = 0.5f * k1;
    float t2_2 = t1_1 * t1_1;
    float t2_3 = 2.0f * (t1_1 * t1_2);
    float t2_4 = t1_2 * t1_2;
    float t3_4 = t2_2 * t1_2 + t2_3 * t1_1;
    float t3_6 = t2_4 * t1_2;
    float t4_4 = t2_2 * t2_2;
    float t4_5 = 2.0f * (t2_2 * t2_3);
    float t4_6 = 2.0f * (t2_2 * t2_4) + t2_3 * t2_3;
    float t4_7 = 2.0f * (t2_3 * t2_4);
    float t4_8 = t2_4 * t2_4;
    float t5_6 = t4_4 * t1_2 + t4_5 * t1_1;
    float t5_8 = t4_6 * t1_2 + t4_7 * t1_1;
    float t6_6 = t4_4 * t2_2;
    float t6_7 = t4_4 * t2_3 + t4_5 * t2_2;
    float t6_8 = t4_4 * t2_4 + t4_5 * t2_3 + t4_6 * t2_2;
    float t7_8 = t6_6 * t1_2 + t6_7 * t1_1;
    float t8_8 = t6_6 * t2_2;
    float u = 1.0f;
    u -= (1.0f / 24.0f) * t2_2 + (1.0f / 160.0f) * t2_4;
    u += (1.0f / 1920.0f) * t4_4 + (1.0f / 10752.0f) * t4_6 + (1.0f / 55296.0f) * t4_8;
    u -= (1.0f / 322560.0f) * t6_6 + (1.0f / 1658880.0f) * t6_8;
    u += (1.0f / 92897280.0f) * t8_8;
    float v = (1.0f / 12.0f) * t1_2;
    v -= (1.0f / 480.0f) * t3_4 + (1.0f / 2688.0f) * t3_6;
    v += (1.0f / 53760.0f) * t5_6 + (1.0f / 276480.0f) * t5_8;
    v -= (1.0f / 11612160.0f) * t7_8;
    return v2(u, v);
}

/* flatten.wgsl:198-216 */
static vec2 es_params_eval(const euler_params *p, float t) {
    float thm = es_params_eval_th(p, t * 0.5f);
    float k0 = p->k0, k1 = p->k1;
    vec2 uv = integ_euler_10((k0 + k1 * (0.5f * t - 0.5f)) * t, k1 * t * t);
    float scale = t / p->ch;
    float s = scale * vo_sinf(thm);
    float cs = scale * vo_cosf(thm);
    float x = uv.x * cs - uv.y * s;
    float y = -uv.y * cs - uv.x * s;
    return v2(x, y);
}
static vec2 es_params_eval_with_offset(const euler_params *p, float t, float offset) {
    float th = es_params_eval_th(p, t);
    vec2 v = v2(offset * vo_sinf(th), offset * vo_cosf(th));
    return vadd(es_params_eval(p, t), v);
}
/* flatten.wgsl:222-227 */
static vec2 es_seg_eval_with_offset(vec2 p0, vec2 p1, const euler_params *p, float t, float normalized_offset) {
    vec2 chord = vsub(p1, p0);
    vec2 xy = es_params_eval_with_offset(p, t, normalized_offset);
    return v2(p0.x + (chord.x * xy.x - chord.y * xy.y), p0.y + (chord.x * xy.y + chord.y * xy.x));
}

static float pow_1_5_signed(float x) { return x * sqrtf(fabsf(x)); }

#define BREAK1 0.8f
#define BREAK2 1.25f
#define BREAK3 2.1f
#define SIN_SCALE 1.0976991822760038f
#define QUAD_A1 0.6406f
#define QUAD_B1 (-0.81f)
#define QUAD_C1 0.9148117935952064f
#define QUAD_A2 0.5f
#define QUAD_B2 (-0.156f)
#define QUAD_C2 0.16145779359520596f
#define QUAD_W1 (0.5f * QUAD_B1 / QUAD_A1)
#define QUAD_V1 (1.0f / QUAD_A1)
#define QUAD_U1 (QUAD_W1 * QUAD_W1 - QUAD_C1 / QUAD_A1)
#define QUAD_W2 (0.5f * QUAD_B2 / QUAD_A2)
#define QUAD_V2 (1.0f / QUAD_A2)
#define QUAD_U2 (QUAD_W2 * QUAD_W2 - QUAD_C2 / QUAD_A2)
#define FRAC_PI_4 0.7853981633974483f
#define CBRT_9_8 1.040041911525952f
#define SQRT8_OVER_3 0.9428090415820634f

/* flatten.wgsl:254-266 */
static float espc_int_approx(float x) {
    float y = fabsf(x);
    float a;
    if (y < BREAK1) {
        a = vo_sinf(SIN_SCALE * y) * (1.0f / SIN_SCALE);
    } else if (y < BREAK2) {
        a = SQRT8_OVER_3 * pow_1_5_signed(y - 1.0f) + FRAC_PI_4;
    } else {
        float qa = y < BREAK3 ? QUAD_A1 : QUAD_A2;
        float qb = y < BREAK3 ? QUAD_B1 : QUAD_B2;
        float qc = y < BREAK3 ? QUAD_C1 : QUAD_C2;
        a = (qa * y + qb) * y + qc;
    }
    return a * vo_sign(x);
}

/* flatten.wgsl:268-282 */
static float espc_int_inv_approx(float x) {
    float y = fabsf(x);
    float a;
    if (y < 0.7010707591262915f) {
        a = vo_asinf(y * SIN_SCALE) * (1.0f / SIN_SCALE);
    } else if (y < 0.903249293595206f) {
        float b = y - FRAC_PI_4;
        float u = vo_powf(fabsf(b), 2.0f / 3.0f) * vo_sign(b);
        a = u * CBRT_9_8 + 1.0f;
    } else {
        int lo = y < 2.038857793595206f;
        float u = lo ? QUAD_U1 : QUAD_U2;
        float v = lo ? QUAD_V1 : QUAD_V2;
        float w = lo ? QUAD_W1 : QUAD_W2;
        a = sqrtf(u + v * y) - w;
    }
    return a * vo_sign(x);
}

/* flatten.wgsl:289-297 */
static point_deriv eval_cubic_and_deriv(vec2 p0, vec2 p1, vec2 p2, vec2 p3, float t) {
    point_deriv r;
    float m = 1.0f - t;
    float mm = m * m;
    float mt = m * t;
    float tt = t * t;
    /* p = p0*(mm*m) + (p1*(3mm) + p2*(3mt) + p3*tt)*t */
    vec2 inner = vadd(vadd(vmul(p1, 3.0f * mm), vmul(p2, 3.0f * mt)), vmul(p3, tt));
    r.point = vadd(vmul(p0, mm * m), vmul(inner, t));
    r.deriv = vadd(vadd(vmul(vsub(p1, p0), mm), vmul(vsub(p2, p1), 2.0f * mt)), vmul(vsub(p3, p2), tt));
    return r;
}

/* flatten.wgsl:299-313 (EPS = 1e-12 on dot) */
static vec2 cubic_start_tangent(vec2 p0, vec2 p1, vec2 p2, vec2 p3) {
    const float EPS = 1e-12f;
    vec2 d01 = vsub(p1, p0), d02 = vsub(p2, p0), d03 = vsub(p3, p0);
    if (vdot(d01, d01) > EPS) return d01;
    if (vdot(d02, d02) > EPS) return d02;
    return d03;
}
static vec2 cubic_end_tangent(vec2 p0, vec2 p1, vec2 p2, vec2 p3) {
    const float EPS = 1e-12f;
    vec2 d23 = vsub(p3, p2), d13 = vsub(p3, p1), d03 = vsub(p3, p0);
    if (vdot(d23, d23) > EPS) return d23;
    if (vdot(d13, d13) > EPS) return d13;
    return d03;
}

enum { ESPC_ROBUST_NORMAL = 0, ESPC_ROBUST_LOW_K1 = 1, ESPC_ROBUST_LOW_DIST = 2 };

/* flatten.wgsl:328-481 */
static void flatten_euler(flat_state *s, const cubic_points *cubic, uint32_t path_ix,
                          const vo_xform *local_to_device, float offset, vec2 start_p, vec2 end_p) {
    vec2 p0, p1, p2, p3;
    float scale;
    vo_xform transform;
    vec2 t_start = start_p, t_end = end_p;
    if (offset == 0.0f) {
        p0 = xf_apply(local_to_device, cubic->p0);
        p1 = xf_apply(local_to_device, cubic->p1);
        p2 = xf_apply(local_to_device, cubic->p2);
        p3 = xf_apply(local_to_device, cubic->p3);
        scale = 1.0f;
        transform.m[0] = 1.0f; transform.m[1] = 0.0f; transform.m[2] = 0.0f; transform.m[3] = 1.0f;
        transform.t[0] = 0.0f; transform.t[1] = 0.0f;
        t_start = p0;
        t_end = p3;
    } else {
        p0 = cubic->p0; p1 = cubic->p1; p2 = cubic->p2; p3 = cubic->p3;
        transform = *local_to_device;
        const float *mat = transform.m;
        scale = 0.5f * (vlen(v2(mat[0] + mat[3], mat[1] - mat[2])) + vlen(v2(mat[0] - mat[3], mat[1] + mat[2])));
    }
    if (p0.x == p1.x && p0.y == p1.y && p0.x == p2.x && p0.y == p2.y && p0.x == p3.x && p0.y == p3.y) {
        return;
    }
    const float tol = 0.25f;
    uint32_t t0_u = 0u;
    float dt = 1.0f;
    vec2 last_p = p0;
    vec2 last_q = vsub(p1, p0);
    if (vdot(last_q, last_q) < DERIV_THRESH_SQUARED) {
        last_q = eval_cubic_and_deriv(p0, p1, p2, p3, DERIV_EPS).deriv;
    }
    float last_t = 0.0f;
    vec2 lp0 = t_start;
    for (;;) {
        float t0 = (float)t0_u * dt;
        if (t0 == 1.0f) break;
        float t1 = t0 + dt;
        vec2 this_p0 = last_p;
        vec2 this_q0 = last_q;
        point_deriv this_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1);
        if (vdot(this_pq1.deriv, this_pq1.deriv) < DERIV_THRESH_SQUARED) {
            point_deriv new_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1 - DERIV_EPS);
            this_pq1.deriv = new_pq1.deriv;
            if (t1 < 1.0f) {
                this_pq1.point = new_pq1.point;
                t1 = t1 - DERIV_EPS;
            }
        }
        float actual_dt = t1 - last_t;
        cubic_params cp = cubic_from_points_derivs(this_p0, this_pq1.point, this_q0, this_pq1.deriv, actual_dt);
        if (cp.err * scale <= tol || dt <= SUBDIV_LIMIT) {
            euler_params ep = es_params_from_angles(cp.th0, cp.th1);
            vec2 es_p0 = this_p0, es_p1 = this_pq1.point;
            float k0 = ep.k0 - 0.5f * ep.k1;
            float k1 = ep.k1;
            float normalized_offset = offset / cp.chord_len;
            float dist_scaled = normalized_offset * ep.ch;
            float scale_multiplier = sqrtf(0.125f * scale * cp.chord_len / (ep.ch * tol));
            float a = 0.0f, b = 0.0f, integral = 0.0f, int0 = 0.0f, n_frac;
            int robust = ESPC_ROBUST_NORMAL;
            if (fabsf(k1) < K1_THRESH) {
                float k = ep.k0;
                n_frac = sqrtf(fabsf(k * (k * dist_scaled + 1.0f)));
                robust = ESPC_ROBUST_LOW_K1;
            } else if (fabsf(dist_scaled) < DIST_THRESH) {
                a = k1;
                b = k0;
                int0 = pow_1_5_signed(b);
                float int1 = pow_1_5_signed(a + b);
                integral = int1 - int0;
                n_frac = (2.0f / 3.0f) * integral / a;
                robust = ESPC_ROBUST_LOW_DIST;
            } else {
                a = -2.0f * dist_scaled * k1;
                b = -1.0f - 2.0f * dist_scaled * k0;
                int0 = espc_int_approx(b);
                float int1 = espc_int_approx(a + b);
                integral = int1 - int0;
                float k_peak = k0 - k1 * b / a;
                float integrand_peak = sqrtf(fabsf(k_peak * (k_peak * dist_scaled + 1.0f)));
                n_frac = integral * integrand_peak / a;
            }
            float n = vo_clamp(ceilf(n_frac * scale_multiplier), 1.0f, 100.0f);
            uint32_t n_u = f2u(n);
            for (uint32_t i = 0; i < n_u; i++) {
                vec2 lp1;
                if (i + 1u == n_u && t1 == 1.0f) {
                    lp1 = t_end;
                } else {
                    float t = (float)(i + 1u) / n;
                    float sv = t;
                    if (robust != ESPC_ROBUST_LOW_K1) {
                        float u = integral * t + int0;
                        float inv;
                        if (robust == ESPC_ROBUST_LOW_DIST) {
                            inv = vo_powf(fabsf(u), 2.0f / 3.0f) * vo_sign(u);
                        } else {
                            inv = espc_int_inv_approx(u);
                        }
                        sv = (inv - b) / a;
                    }
                    lp1 = es_seg_eval_with_offset(es_p0, es_p1, &ep, sv, normalized_offset);
                }
                vec2 l0 = offset >= 0.0f ? lp0 : lp1;
                vec2 l1 = offset >= 0.0f ? lp1 : lp0;
                output_line_xf(s, path_ix, l0, l1, &transform);
                lp0 = lp1;
            }
            last_p = this_pq1.point;
            last_q = this_pq1.deriv;
            last_t = t1;
            t0_u += 1u;
            uint32_t shift = (uint32_t)__builtin_ctz(t0_u);
            t0_u >>= shift;
            dt *= (float)(1u << shift);
        } else {
            t0_u = t0_u * 2u;
            dt *= 0.5f;
        }
    }
}

/* flatten.wgsl:494-521 */
static void flatten_arc(flat_state *s, uint32_t path_ix, vec2 begin, vec2 end, vec2 center, float angle, const vo_xform *transform) {
    vec2 p0 = xf_apply(transform, begin);
    vec2 r = vsub(begin, center);
    const float MIN_THETA = 0.0001f;
    const float tol = 0.25f;
    float radius = vo_max(tol, vlen(vsub(p0, xf_apply(transform, center))));
    float theta = vo_max(MIN_THETA, 2.0f * vo_acosf(1.0f - tol / radius));
    uint32_t n_lines = umax(1u, f2u(ceilf(angle / theta)));
    float cs = vo_cosf(theta);
    float sn = vo_sinf(theta);
    uint32_t line_ix = alloc_lines(s, n_lines);
    for (uint32_t i = 0; i < n_lines - 1u; i++) {
        /* rot = mat2x2(c, -s, s, c) (column major) */
        r = v2(cs * r.x + sn * r.y, -sn * r.x + cs * r.y);
        vec2 p1 = xf_apply(transform, vadd(center, r));
        write_line(s, line_ix + i, path_ix, p0, p1);
        p0 = p1;
    }
    vec2 p1 = xf_apply(transform, end);
    write_line(s, line_ix + n_lines - 1u, path_ix, p0, p1);
}

/* flatten.wgsl:523-547 */
static void draw_cap(flat_state *s, uint32_t path_ix, uint32_t cap_style, vec2 point, vec2 cap0, vec2 cap1,
                     vec2 offset_tangent, const vo_xform *transform) {
    if (cap_style == STYLE_FLAGS_CAP_ROUND) {
        flatten_arc(s, path_ix, cap0, cap1, point, 3.1415927f, transform);
        return;
    }
    vec2 start = cap0, end = cap1;
    int is_square = cap_style == STYLE_FLAGS_CAP_SQUARE;
    uint32_t line_ix = alloc_lines(s, is_square ? 3u : 1u);
    if (is_square) {
        vec2 v = offset_tangent;
        vec2 p0 = vadd(start, v);
        vec2 p1 = vadd(end, v);
        write_line_xf(s, line_ix + 1u, path_ix, start, p0, transform);
        write_line_xf(s, line_ix + 2u, path_ix, p1, end, transform);
        start = p0;
        end = p1;
    }
    write_line_xf(s, line_ix, path_ix, start, end, transform);
}

/* vello_encoding/src/math.rs:127-150 (== unpack2x16float()[0]) */
static float f16_to_f32(uint32_t bits) {
    const uint32_t MAGIC = 113u << 23;
    const uint32_t SHIFTED_EXP = 0x7c00u << 13;
    uint32_t o = (bits & 0x7fffu) << 13;
    uint32_t e = SHIFTED_EXP & o;
    o += (127u - 15u) << 23;
    if (e == SHIFTED_EXP) {
        o += (128u - 16u) << 23;
    } else if (e == 0u) {
        o += 1u << 23;
        o = f2bits(bits2f(o) - bits2f(MAGIC));
    }
    return bits2f(o | ((bits & 0x8000u) << 16));
}

/* flatten.wgsl:549-631 */
static void draw_join(flat_state *s, uint32_t path_ix, uint32_t style_flags, vec2 p0, vec2 tan_prev, vec2 tan_next,
                      vec2 n_prev, vec2 n_next, const vo_xform *transform) {
    vec2 front0 = vadd(p0, n_prev);
    vec2 front1 = vadd(p0, n_next);
    vec2 back0 = vsub(p0, n_next);
    vec2 back1 = vsub(p0, n_prev);
    float cr = tan_prev.x * tan_next.y - tan_prev.y * tan_next.x;
    float d = vdot(tan_prev, tan_next);
    switch (style_flags & STYLE_FLAGS_JOIN_MASK) {
    case STYLE_FLAGS_JOIN_BEVEL:
        output_two_lines_xf(s, path_ix, front0, front1, back0, back1, transform);
        break;
    case STYLE_FLAGS_JOIN_MITER: {
        float hyp = vlen(v2(cr, d));
        float miter_limit = f16_to_f32(style_flags & STYLE_MITER_LIMIT_MASK);
        uint32_t line_ix;
        if (2.0f * hyp < (hyp + d) * miter_limit * miter_limit && fabsf(cr) > TANGENT_THRESH * TANGENT_THRESH) {
            int is_backside = cr > 0.0f;
            vec2 fp_last = is_backside ? back1 : front0;
            vec2 fp_this = is_backside ? back0 : front1;
            vec2 p = is_backside ? back0 : front0;
            vec2 v = vsub(fp_this, fp_last);
            float h = (tan_prev.x * v.y - tan_prev.y * v.x) / cr;
            vec2 miter_pt = vsub(fp_this, vmul(tan_next, h));
            line_ix = alloc_lines(s, 3u);
            write_line_xf(s, line_ix, path_ix, p, miter_pt, transform);
            line_ix += 1u;
            if (is_backside) back0 = miter_pt; else front0 = miter_pt;
        } else {
            line_ix = alloc_lines(s, 2u);
        }
        write_line_xf(s, line_ix, path_ix, front0, front1, transform);
        write_line_xf(s, line_ix + 1u, path_ix, back0, back1, transform);
        break;
    }
    case STYLE_FLAGS_JOIN_ROUND: {
        vec2 arc0, arc1, other0, other1;
        if (cr > 0.0f) { arc0 = back0; arc1 = back1; other0 = front0; other1 = front1; }
        else { arc0 = front0; arc1 = front1; other0 = back0; other1 = back1; }
        flatten_arc(s, path_ix, arc0, arc1, p0, fabsf(vo_atan2f(cr, d)), transform);
        output_line_xf(s, path_ix, other0, other1, transform);
        break;
    }
    default: break;
    }
}

typedef struct { uint32_t tag_byte; vo_tag_monoid monoid; } path_tag_data;

/* flatten.wgsl:684-701 */
static path_tag_data compute_tag_monoid(const vo_ctx *c, uint32_t ix) {
    const vo_tag_monoid *tag_monoids = (const vo_tag_monoid *)c->buf[VO_BUF_TAG_MONOIDS];
    uint32_t tag_word = c->scene[c->cfg.layout.path_tag_base + (ix >> 2)];
    uint32_t shift = (ix & 3u) * 8u;
    vo_tag_monoid tm = vo_reduce_tag(tag_word & ((1u << shift) - 1u));
    tm = combine_tag(tag_monoids[ix >> 2], tm);
    path_tag_data r;
    r.tag_byte = (tag_word >> shift) & 0xffu;
    tm.trans_ix -= 1u;
    tm.style_ix -= STYLE_SIZE_IN_WORDS;
    r.monoid = tm;
    return r;
}

static vec2 read_f32_point(const vo_ctx *c, uint32_t ix) {
    const uint32_t *pd = c->scene + c->cfg.layout.path_data_base;
    return v2(bits2f(pd[ix]), bits2f(pd[ix + 1u]));
}
static vec2 read_i16_point(const vo_ctx *c, uint32_t ix) {
    uint32_t raw = c->scene[c->cfg.layout.path_data_base + ix];
    float x = (float)(((int32_t)(raw << 16)) >> 16);
    float y = (float)(((int32_t)raw) >> 16);
    return v2(x, y);
}

/* flatten.wgsl:710-764 */
static cubic_points read_path_segment(const vo_ctx *c, const path_tag_data *tag, int is_stroke) {
    vec2 p0, p1, p2 = v2(0, 0), p3 = v2(0, 0);
    uint32_t seg_type = tag->tag_byte & PATH_TAG_SEG_TYPE;
    uint32_t off = tag->monoid.pathseg_offset;
    int is_stroke_cap_marker = is_stroke && (tag->tag_byte & PATH_TAG_SUBPATH_END) != 0u;
    int is_open = seg_type == PATH_TAG_QUADTO;
    if ((tag->tag_byte & PATH_TAG_F32) != 0u) {
        p0 = read_f32_point(c, off);
        p1 = read_f32_point(c, off + 2u);
        if (seg_type >= PATH_TAG_QUADTO) {
            p2 = read_f32_point(c, off + 4u);
            if (seg_type == PATH_TAG_CUBICTO) p3 = read_f32_point(c, off + 6u);
        }
    } else {
        p0 = read_i16_point(c, off);
        p1 = read_i16_point(c, off + 1u);
        if (seg_type >= PATH_TAG_QUADTO) {
            p2 = read_i16_point(c, off + 2u);
            if (seg_type == PATH_TAG_CUBICTO) p3 = read_i16_point(c, off + 3u);
        }
    }
    if (is_stroke_cap_marker && is_open) {
        p0 = p1;
        p1 = p2;
        seg_type = PATH_TAG_LINETO;
    }
    /* degree raise: p + (1/3)*(q - p) */
    const float third = 1.0f / 3.0f;
    if (seg_type == PATH_TAG_LINETO) {
        p3 = p1;
        p2 = vadd(p3, vmul(vsub(p0, p3), third));
        p1 = vadd(p0, vmul(vsub(p3, p0), third));
    } else if (seg_type == PATH_TAG_QUADTO) {
        p3 = p2;
        p2 = vadd(p1, vmul(vsub(p2, p1), third));
        p1 = vadd(p1, vmul(vsub(p0, p1), third));
    }
    cubic_points r = {p0, p1, p2, p3};
    return r;
}

/* flatten.wgsl:831-923 */
void vo_stage_flatten(vo_ctx *c) {
    vo_path_bbox *path_bboxes = (vo_path_bbox *)c->buf[VO_BUF_PATH_BBOXES];
    const vo_layout *L = &c->cfg.layout;
    uint32_t n_tags = c->n_tag_words * 4u;
    /* one invocation per tag, as the shader: independent but for the line bump and the per-path bbox atomics
     * (flatten.wgsl:916-921), so the CPU-baseline mode simply runs the tags on n_threads threads */
#pragma omp parallel for schedule(dynamic, 4096) if (c->n_threads > 1) num_threads(VO_OMP_THREADS(c))
    for (uint32_t ix = 0; ix < n_tags; ix++) {
        flat_state st;
        st.c = c;
        st.lines = (vo_line_soup *)c->buf[VO_BUF_LINES];
        st.bump = (vo_bump *)c->buf[VO_BUF_BUMP];
        st.bbox[0] = 1e31f; st.bbox[1] = 1e31f; st.bbox[2] = -1e31f; st.bbox[3] = -1e31f;
        path_tag_data tag = compute_tag_monoid(c, ix);
        uint32_t path_ix = tag.monoid.path_ix;
        uint32_t style_ix = tag.monoid.style_ix;
        uint32_t trans_ix = tag.monoid.trans_ix;
        uint32_t seg_type = tag.tag_byte & PATH_TAG_SEG_TYPE;
        if ((tag.tag_byte & PATH_TAG_PATH) == 0u && seg_type == 0u) continue; /* nothing observable */
        uint32_t style_flags = c->scene[(uint32_t)(L->style_base + style_ix)];
        uint32_t draw_flags = (style_flags & STYLE_FLAGS_FILL) == 0u ? 0u : DRAW_INFO_FLAGS_FILL_RULE_BIT;
        /* resolve appends one PATH marker per unclosed layer behind the last counted path (resolve.rs:127-129): their
         * stores land outside path_bboxes[n_paths], where WebGPU's robust buffer access drops them */
        if ((tag.tag_byte & PATH_TAG_PATH) != 0u && path_ix < L->n_paths) {
            path_bboxes[path_ix].draw_flags = draw_flags;
            path_bboxes[path_ix].trans_ix = trans_ix;
        }
        if (seg_type != 0u) {
            int is_stroke = (style_flags & STYLE_FLAGS_STYLE) != 0u;
            vo_xform transform = vo_read_transform(c->scene, L->transform_base, trans_ix);
            cubic_points pts = read_path_segment(c, &tag, is_stroke);
            if (is_stroke) {
                float linewidth = bits2f(c->scene[L->style_base + style_ix + 1u]);
                float offset = 0.5f * linewidth;
                int is_open = seg_type != PATH_TAG_LINETO;
                int is_stroke_cap_marker = (tag.tag_byte & PATH_TAG_SUBPATH_END) != 0u;
                if (is_stroke_cap_marker) {
                    if (is_open) {
                        vec2 tangent = vsub(pts.p3, pts.p0);
                        vec2 offset_tangent = vmul(vnorm(tangent), offset);
                        vec2 n = v2(-offset_tangent.y, offset_tangent.x);
                        draw_cap(&st, path_ix, (style_flags & STYLE_FLAGS_START_CAP_MASK) >> 2, pts.p0,
                                 vsub(pts.p0, n), vadd(pts.p0, n), vneg(offset_tangent), &transform);
                    }
                } else {
                    /* read_neighboring_segment(ix + 1), flatten.wgsl:810-822 */
                    path_tag_data ntag = compute_tag_monoid(c, ix + 1u);
                    cubic_points npts = read_path_segment(c, &ntag, 1);
                    int n_is_closed = (ntag.tag_byte & PATH_TAG_SEG_TYPE) == PATH_TAG_LINETO;
                    int n_is_marker = (ntag.tag_byte & PATH_TAG_SUBPATH_END) != 0u;
                    int do_join = !n_is_marker || n_is_closed;
                    vec2 n_tangent = vsub(npts.p3, npts.p0);
                    if (!n_is_marker) n_tangent = cubic_start_tangent(npts.p0, npts.p1, npts.p2, npts.p3);

                    vec2 tan_start = cubic_start_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                    if (vdot(tan_start, tan_start) < TANGENT_THRESH * TANGENT_THRESH) tan_start = v2(TANGENT_THRESH, 0.0f);
                    vec2 tan_prev = cubic_end_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                    if (vdot(tan_prev, tan_prev) < TANGENT_THRESH * TANGENT_THRESH) tan_prev = v2(TANGENT_THRESH, 0.0f);
                    vec2 tan_next = n_tangent;
                    if (vdot(tan_next, tan_next) < TANGENT_THRESH * TANGENT_THRESH) tan_next = v2(TANGENT_THRESH, 0.0f);
                    vec2 n_start = vmul(vnorm(v2(-tan_start.y, tan_start.x)), offset);
                    vec2 offset_tangent = vmul(vnorm(tan_prev), offset);
                    vec2 n_prev = v2(-offset_tangent.y, offset_tangent.x);
                    vec2 tnn = vmul(vnorm(tan_next), offset);
                    vec2 n_next = v2(-tnn.y, tnn.x);

                    flatten_euler(&st, &pts, path_ix, &transform, offset, vadd(pts.p0, n_start), vadd(pts.p3, n_prev));
                    flatten_euler(&st, &pts, path_ix, &transform, -offset, vsub(pts.p0, n_start), vsub(pts.p3, n_prev));
                    if (do_join) {
                        draw_join(&st, path_ix, style_flags, pts.p3, tan_prev, tan_next, n_prev, n_next, &transform);
                    } else {
                        draw_cap(&st, path_ix, style_flags & STYLE_FLAGS_END_CAP_MASK, pts.p3, vadd(pts.p3, n_prev),
                                 vsub(pts.p3, n_prev), offset_tangent, &transform);
                    }
                }
            } else {
                flatten_euler(&st, &pts, path_ix, &transform, 0.0f, pts.p0, pts.p3);
            }
            if ((st.bbox[2] > st.bbox[0] || st.bbox[3] > st.bbox[1]) && path_ix < L->n_paths) {
                vo_path_bbox *out = &path_bboxes[path_ix];
                atomic_min_i32(&out->x0, f2i(floorf(st.bbox[0])));
                atomic_min_i32(&out->y0, f2i(floorf(st.bbox[1])));
                atomic_max_i32(&out->x1, f2i(ceilf(st.bbox[2])));
                atomic_max_i32(&out->y1, f2i(ceilf(st.bbox[3])));
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* draw_reduce + draw_leaf: cpu/draw_reduce.rs, cpu/draw_leaf.rs,      */
/* shader/draw_leaf.wgsl:52-289                                        */
/* ------------------------------------------------------------------ */
static vo_xform xf_inverse(const vo_xform *t) {
    /* shared/transform.wgsl:13-18 */
    vo_xform r;
    float inv_det = 1.0f / (t->m[0] * t->m[3] - t->m[1] * t->m[2]);
    r.m[0] = inv_det * t->m[3];
    r.m[1] = inv_det * -t->m[1];
    r.m[2] = inv_det * -t->m[2];
    r.m[3] = inv_det * t->m[0];
    /* mat2x2(inv.xy, inv.zw) * -translate */
    float tx = -t->t[0], ty = -t->t[1];
    r.t[0] = r.m[0] * tx + r.m[2] * ty;
    r.t[1] = r.m[1] * tx + r.m[3] * ty;
    return r;
}
static vo_xform xf_mul(const vo_xform *a, const vo_xform *b) {
    /* shared/transform.wgsl:20-25 */
    vo_xform r;
    r.m[0] = a->m[0] * b->m[0] + a->m[2] * b->m[1];
    r.m[1] = a->m[1] * b->m[0] + a->m[3] * b->m[1];
    r.m[2] = a->m[0] * b->m[2] + a->m[2] * b->m[3];
    r.m[3] = a->m[1] * b->m[2] + a->m[3] * b->m[3];
    r.t[0] = a->m[0] * b->t[0] + a->m[2] * b->t[1] + a->t[0];
    r.t[1] = a->m[1] * b->t[0] + a->m[3] * b->t[1] + a->t[1];
    return r;
}
static vec2 xf_apply_plain(const vo_xform *t, vec2 p) {
    /* shared/transform.wgsl:9-11 */
    return v2(t->m[0] * p.x + t->m[2] * p.y + t->t[0], t->m[1] * p.x + t->m[3] * p.y + t->t[1]);
}
static vo_xform from_poly2(vec2 p0, vec2 p1) {
    vo_xform r;
    r.m[0] = p1.y - p0.y; r.m[1] = p0.x - p1.x; r.m[2] = p1.x - p0.x; r.m[3] = p1.y - p0.y;
    r.t[0] = p0.x; r.t[1] = p0.y;
    return r;
}
static vo_xform two_point_to_unit_line(vec2 p0, vec2 p1) {
    vo_xform tmp1 = from_poly2(p0, p1);
    vo_xform inv = xf_inverse(&tmp1);
    vo_xform tmp2 = from_poly2(v2(0, 0), v2(1.0f, 0.0f));
    return xf_mul(&tmp2, &inv);
}
static void write_xform(uint32_t *info, const vo_xform *x) {
    for (int i = 0; i < 4; i++) info[i] = f2bits(x->m[i]);
    info[4] = f2bits(x->t[0]);
    info[5] = f2bits(x->t[1]);
}

void vo_stage_draw_scan(vo_ctx *c) {
    const vo_layout *L = &c->cfg.layout;
    const uint32_t *scene = c->scene;
    vo_draw_monoid *draw_monoid = (vo_draw_monoid *)c->buf[VO_BUF_DRAW_MONOIDS];
    uint32_t *info = (uint32_t *)c->buf[VO_BUF_INFO_BIN_DATA];
    vo_clip *clip_inp = (vo_clip *)c->buf[VO_BUF_CLIP_INP];
    const vo_path_bbox *path_bbox = (const vo_path_bbox *)c->buf[VO_BUF_PATH_BBOXES];
    vo_draw_monoid m = {0, 0, 0, 0};
    for (uint32_t ix = 0; ix < L->n_draw_objects; ix++) {
        uint32_t tag_word = scene[L->draw_tag_base + ix];
        draw_monoid[ix] = m;
        uint32_t dd = L->draw_data_base + m.scene_offset;
        uint32_t di = m.info_offset;
        if (tag_word == DRAWTAG_FILL_COLOR || tag_word == DRAWTAG_FILL_LIN_GRADIENT ||
            tag_word == DRAWTAG_FILL_RAD_GRADIENT || tag_word == DRAWTAG_FILL_SWEEP_GRADIENT ||
            tag_word == DRAWTAG_FILL_IMAGE || tag_word == DRAWTAG_BEGIN_CLIP || tag_word == DRAWTAG_BLURRED_ROUNDED_RECT) {
            vo_path_bbox bbox = path_bbox[m.path_ix];
            uint32_t draw_flags = bbox.draw_flags;
            vo_xform transform;
            memset(&transform, 0, sizeof transform);
            if (tag_word != DRAWTAG_FILL_COLOR && tag_word != DRAWTAG_BEGIN_CLIP)
                transform = vo_read_transform(scene, L->transform_base, bbox.trans_ix);
            switch (tag_word) {
            case DRAWTAG_FILL_COLOR:
            case DRAWTAG_BEGIN_CLIP:
                info[di] = draw_flags;
                break;
            case DRAWTAG_FILL_LIN_GRADIENT: {
                /* draw_leaf.wgsl:137-150 */
                info[di] = draw_flags;
                vec2 p0 = v2(bits2f(scene[dd + 1]), bits2f(scene[dd + 2]));
                vec2 p1 = v2(bits2f(scene[dd + 3]), bits2f(scene[dd + 4]));
                p0 = xf_apply_plain(&transform, p0);
                p1 = xf_apply_plain(&transform, p1);
                vec2 dxy = vsub(p1, p0);
                float scale = 1.0f / vdot(dxy, dxy);
                vec2 line_xy = vmul(dxy, scale);
                float line_c = -vdot(p0, line_xy);
                info[di + 1] = f2bits(line_xy.x);
                info[di + 2] = f2bits(line_xy.y);
                info[di + 3] = f2bits(line_c);
                break;
            }
            case DRAWTAG_FILL_RAD_GRADIENT: {
                /* draw_leaf.wgsl:151-226 */
                const float GRADIENT_EPSILON = 1.0f / (float)(1 << 12);
                info[di] = draw_flags;
                vec2 p0 = v2(bits2f(scene[dd + 1]), bits2f(scene[dd + 2]));
                vec2 p1 = v2(bits2f(scene[dd + 3]), bits2f(scene[dd + 4]));
                float r0 = bits2f(scene[dd + 5]);
                float r1 = bits2f(scene[dd + 6]);
                vo_xform user_to_gradient = xf_inverse(&transform);
                vo_xform xform;
                float focal_x = 0.0f, radius;
                uint32_t kind, flags = 0u;
                if (fabsf(r0 - r1) < GRADIENT_EPSILON) {
                    kind = RAD_GRAD_KIND_STRIP;
                    float scaled = r0 / vlen(vsub(p0, p1));
                    vo_xform u = two_point_to_unit_line(p0, p1);
                    xform = xf_mul(&u, &user_to_gradient);
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
                    vec2 cf = vadd(vmul(p0, 1.0f - focal_x), vmul(p1, focal_x));
                    radius = r1 / vlen(vsub(cf, p1));
                    vo_xform u = two_point_to_unit_line(cf, p1);
                    vo_xform user_to_unit_line = xf_mul(&u, &user_to_gradient);
                    vo_xform sc;
                    memset(&sc, 0, sizeof sc);
                    if (fabsf(radius - 1.0f) <= GRADIENT_EPSILON) {
                        kind = RAD_GRAD_KIND_FOCAL_ON_CIRCLE;
                        float scale = 0.5f * fabsf(1.0f - focal_x);
                        sc.m[0] = scale; sc.m[3] = scale;
                    } else {
                        float a = radius * radius - 1.0f;
                        float scale_ratio = fabsf(1.0f - focal_x) / a;
                        sc.m[0] = radius * scale_ratio;
                        sc.m[3] = sqrtf(fabsf(a)) * scale_ratio;
                    }
                    xform = xf_mul(&sc, &user_to_unit_line);
                }
                write_xform(info + di + 1, &xform);
                info[di + 7] = f2bits(focal_x);
                info[di + 8] = f2bits(radius);
                info[di + 9] = (flags << 3) | kind;
                break;
            }
            case DRAWTAG_FILL_SWEEP_GRADIENT: {
                info[di] = draw_flags;
                vec2 p0 = v2(bits2f(scene[dd + 1]), bits2f(scene[dd + 2]));
                vo_xform tr;
                tr.m[0] = 1.0f; tr.m[1] = 0.0f; tr.m[2] = 0.0f; tr.m[3] = 1.0f; tr.t[0] = p0.x; tr.t[1] = p0.y;
                vo_xform mm = xf_mul(&transform, &tr);
                vo_xform xform = xf_inverse(&mm);
                write_xform(info + di + 1, &xform);
                info[di + 7] = scene[dd + 3];
                info[di + 8] = scene[dd + 4];
                break;
            }
            case DRAWTAG_FILL_IMAGE: {
                info[di] = draw_flags;
                vo_xform xform = xf_inverse(&transform);
                write_xform(info + di + 1, &xform);
                info[di + 7] = scene[dd];
                info[di + 8] = scene[dd + 1];
                info[di + 9] = scene[dd + 2];
                break;
            }
            case DRAWTAG_BLURRED_ROUNDED_RECT: {
                info[di] = draw_flags;
                vo_xform xform = xf_inverse(&transform);
                write_xform(info + di + 1, &xform);
                info[di + 7] = scene[dd + 1];
                info[di + 8] = scene[dd + 2];
                info[di + 9] = scene[dd + 3];
                info[di + 10] = scene[dd + 4];
                break;
            }
            default: break;
            }
        }
        if (tag_word == DRAWTAG_BEGIN_CLIP || tag_word == DRAWTAG_END_CLIP) {
            uint32_t path_ix = ~ix;
            if (tag_word == DRAWTAG_BEGIN_CLIP) path_ix = m.path_ix;
            clip_inp[m.clip_ix].ix = ix;
            clip_inp[m.clip_ix].path_ix = (int32_t)path_ix;
        }
        /* map_draw_tag + combine (shared/drawtag.wgsl:38-54) */
        m.path_ix += (tag_word != DRAWTAG_NOP) ? 1u : 0u;
        m.clip_ix += tag_word & 1u;
        m.scene_offset += (tag_word >> 2) & 0x07u;
        m.info_offset += (tag_word >> 6) & 0x0fu;
    }
}

/* ------------------------------------------------------------------ */
/* clip_reduce + clip_leaf: cpu/clip_leaf.rs:21-72 (sequential stack;  */
/* equals the partitioned clip_leaf.wgsl:80-217 by construction)       */
/* ------------------------------------------------------------------ */
void vo_stage_clip(vo_ctx *c) {
    uint32_t n_clips = c->cfg.layout.n_clips;
    if (n_clips == 0) return;
    const vo_clip *clip_inp = (const vo_clip *)c->buf[VO_BUF_CLIP_INP];
    const vo_path_bbox *path_bboxes = (const vo_path_bbox *)c->buf[VO_BUF_PATH_BBOXES];
    vo_draw_monoid *draw_monoids = (vo_draw_monoid *)c->buf[VO_BUF_DRAW_MONOIDS];
    float(*clip_bboxes)[4] = (float(*)[4])c->buf[VO_BUF_CLIP_BBOXES];
    typedef struct { uint32_t parent_ix, path_ix; float bbox[4]; } stack_el;
    stack_el *stack = (stack_el *)malloc(sizeof(stack_el) * (n_clips + 1));
    uint32_t sp = 0;
    for (uint32_t gi = 0; gi < n_clips; gi++) {
        vo_clip el = clip_inp[gi];
        if (el.path_ix >= 0) {
            vo_path_bbox pb = path_bboxes[el.path_ix];
            float b[4] = {(float)pb.x0, (float)pb.y0, (float)pb.x1, (float)pb.y1};
            if (sp > 0) {
                const float *l = stack[sp - 1].bbox;
                b[0] = vo_max(b[0], l[0]); b[1] = vo_max(b[1], l[1]);
                b[2] = vo_min(b[2], l[2]); b[3] = vo_min(b[3], l[3]);
            }
            memcpy(clip_bboxes[gi], b, sizeof b);
            stack[sp].parent_ix = el.ix;
            stack[sp].path_ix = (uint32_t)el.path_ix;
            memcpy(stack[sp].bbox, b, sizeof b);
            sp++;
        } else {
            if (sp == 0) continue; /* malformed; resolve() guarantees balance */
            stack_el tos = stack[--sp];
            float b[4] = {-1e9f, -1e9f, 1e9f, 1e9f};
            if (sp > 0) memcpy(b, stack[sp - 1].bbox, sizeof b);
            memcpy(clip_bboxes[gi], b, sizeof b);
            draw_monoids[el.ix].path_ix = tos.path_ix;
            draw_monoids[el.ix].scene_offset = draw_monoids[tos.parent_ix].scene_offset;
            draw_monoids[el.ix].info_offset = draw_monoids[tos.parent_ix].info_offset;
        }
    }
    free(stack);
}
