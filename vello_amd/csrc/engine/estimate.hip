// vello_hip_estimate_capacities: conservative pool sizes for a packed scene BEFORE it is rendered, so that robust mode
// (vello_hip_set_auto_grow) sizes the pools once instead of discovering the demand by failing stage after stage.
//
// Reference: vello_encoding/src/estimate.rs (BumpEstimator, :54-165 count_path, :165-190 tally, :237-250
// estimate_arc_lines, :296-330 segment counts, :360-402 Wang's formula).  Upstream the estimator runs while a Scene is
// built (on kurbo PathEls and a kurbo Stroke) and covers lines and segments only ("TODO: support binning / ptcl / tile",
// estimate.rs:14-16).  The C ABI sees the scene after Resolver::resolve, so the same counting rules are applied to
// the packed path-tag / path-data / transform / style streams (vello_encoding/src/path.rs:246-364), and the three
// TODOs are filled in from the paths' control-point bounding boxes: tiles = the bbox's tile rectangle clipped to the
// target (what tile_alloc allocates), bin entries = the bins that rectangle meets, PTCL words = 7 per (path, tile) pair
// that can hold a command (every tile of a fill's rectangle; a stroke only where it has segments).
// Host code only; no device work.
#include "../../../include/vello_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"

namespace {

using vk::Layout;

constexpr double RSQRT_OF_TOL = 2.2360679775;  // estimate.rs:10, tol = 0.2

struct V2 {
    double x, y;
};
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator*(double s, V2 a) { return {s * a.x, s * a.y}; }
inline double len(V2 a) { return std::sqrt(a.x * a.x + a.y * a.y); }
inline V2 lerp(V2 a, V2 b, double t) { return {a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t}; }

struct Xf {
    double m[6];
};
inline V2 lin(const Xf &t, V2 v) { return {t.m[0] * v.x + t.m[2] * v.y, t.m[1] * v.x + t.m[3] * v.y}; }  // estimate.rs:277-282
inline V2 apply(const Xf &t, V2 v) { return {t.m[0] * v.x + t.m[2] * v.y + t.m[4], t.m[1] * v.x + t.m[3] * v.y + t.m[5]}; }
inline double transform_scale(const Xf &t) {  // estimate.rs:284-296
    double v1x = t.m[0] + t.m[3], v2x = t.m[0] - t.m[3], v1y = t.m[1] - t.m[2], v2y = t.m[1] + t.m[2];
    return std::sqrt(v1x * v1x + v1y * v1y) + std::sqrt(v2x * v2x + v2y * v2y);
}
inline uint32_t sat_u32(double v) { return v <= 0.0 || v != v ? 0u : (v >= 4294967295.0 ? 0xffffffffu : (uint32_t)v); }
inline uint32_t segs_for_line(V2 p0, V2 p1, const Xf &t) {  // estimate.rs:318-323
    V2 d = lin(t, p0 - p1);
    double s = std::ceil(std::ceil(std::fabs(d.x)) * 0.0625) + std::ceil(std::ceil(std::fabs(d.y)) * 0.0625);
    return std::max(sat_u32(s), 1u);
}
inline uint32_t segs_for_len(double w) { return std::max(sat_u32(std::ceil(w * 0.0625 * 1.4142135623730951)), 1u); }  // :326-330
inline double segs_for_cubic(V2 p0, V2 p1, V2 p2, V2 p3, const Xf &t) {  // estimate.rs:298-311
    p0 = lin(t, p0); p1 = lin(t, p1); p2 = lin(t, p2); p3 = lin(t, p3);
    double arc = 0.5 * (len(p3 - p0) + len(p1 - p0) + len(p2 - p1) + len(p3 - p2));
    return std::ceil(arc * 0.0625 * 1.4142135623730951);
}
inline double wang_quad(V2 p0, V2 p1, V2 p2, const Xf &t) {  // estimate.rs:376-381
    V2 v = lin(t, (-2.0 * p1) + p0 + p2);
    return std::ceil(0.5 * std::sqrt(len(v)) * RSQRT_OF_TOL);
}
inline double wang_cubic(V2 p0, V2 p1, V2 p2, V2 p3, const Xf &t) {  // estimate.rs:383-401
    V2 v1 = lin(t, (-2.0 * p1) + p0 + p2), v2 = lin(t, (-2.0 * p2) + p1 + p3);
    return std::ceil(0.86602540378 * std::sqrt(std::max(len(v1), len(v2))) * RSQRT_OF_TOL);
}
inline void arc_lines(double scaled_width, uint32_t &n, double &line_len) {  // estimate.rs:237-250
    const double MIN_THETA = 1e-6, TOL = 0.25;
    double radius = std::max(TOL, scaled_width * 0.5);
    double theta = std::max(2.0 * std::acos(1.0 - TOL / radius), MIN_THETA);
    n = std::max(sat_u32(std::ceil(1.5707963267948966 / theta)), 2u);
    line_len = 2.0 * std::sin(theta) * radius;
}

struct Totals {
    uint64_t linetos = 0, curves = 0, curve_count = 0, segments = 0, tiles = 0, bins = 0, ptcl_pairs = 0;
};

// the per-path accumulators of count_path (estimate.rs:54-75)
struct PathAcc {
    uint32_t caps = 1, joins = 0, lineto_lines = 0, fill_close_lines = 1, curve_lines = 0, curve_count = 0;
    uint64_t segments = 0;
    bool any = false, have_first = false, have_last = false;
    V2 first{0, 0}, last{0, 0};
    double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;  // control-point bbox in target space
    void bbox(V2 p) {
        if (p.x == p.x && p.y == p.y) {
            x0 = std::min(x0, p.x); y0 = std::min(y0, p.y);
            x1 = std::max(x1, p.x); y1 = std::max(y1, p.y);
        }
    }
};

}  // namespace

extern "C" int vello_hip_estimate_capacities(const uint8_t *scene, size_t scene_len, const vello_hip_layout *layout,
                                             const vello_hip_render_params *params, vello_hip_capacities *out) {
    if ((!scene && scene_len) || !layout || !params || !out || (scene_len & 3u)) return VELLO_HIP_E_INVALID;
    const vello_hip_layout &L = *layout;
    const size_t words = scene_len / 4u;
    if (L.path_tag_base > L.path_data_base || L.path_data_base > L.draw_tag_base || L.draw_tag_base > L.draw_data_base ||
        L.draw_data_base > L.transform_base || L.transform_base > L.style_base || L.style_base > words)
        return VELLO_HIP_E_INVALID;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(scene);
    const uint8_t *tags = scene + (size_t)L.path_tag_base * 4u;
    const size_t n_tags = ((size_t)L.path_data_base - L.path_tag_base) * 4u;
    const size_t n_data = (size_t)L.draw_tag_base - L.path_data_base;
    const size_t n_xf = ((size_t)L.style_base - L.transform_base) / 6u, n_style = (words - L.style_base) / 2u;
    const uint32_t wt = (params->width + 15u) / 16u, ht = (params->height + 15u) / 16u;
    const uint32_t wb = (wt + 15u) / 16u, hb = (ht + 15u) / 16u;

    Totals T;
    PathAcc a;
    Xf xf{{1, 0, 0, 1, 0, 0}};
    bool is_stroke = false;
    double width = 0.0;
    uint32_t style_flags = 0u;
    size_t off = 0, n_xf_seen = 0, n_style_seen = 0;  // path-data word offset, markers seen so far

    auto read_pt = [&](size_t o, bool f32, V2 &p) -> bool {
        if (f32) {
            if (o + 2u > n_data) return false;
            float x, y;
            std::memcpy(&x, &w[L.path_data_base + o], 4);
            std::memcpy(&y, &w[L.path_data_base + o + 1u], 4);
            p = {x, y};
        } else {
            if (o + 1u > n_data) return false;
            uint32_t raw = w[L.path_data_base + o];
            p = {(double)(int16_t)(raw & 0xffffu), (double)(int16_t)(raw >> 16)};
        }
        return true;
    };
    auto finish_path = [&]() {
        if (!a.any) {
            a = PathAcc();
            return;
        }
        const double scale = transform_scale(xf);
        const double scaled_width = is_stroke ? width * scale : 0.0;
        uint64_t seg_before = T.segments;
        if (!is_stroke) {  // estimate.rs:139-150
            T.linetos += a.lineto_lines + a.fill_close_lines;
            T.curves += a.curve_lines;
            T.curve_count += a.curve_count;
            T.segments += a.segments;
            if (a.have_first && a.have_last) T.segments += segs_for_line(a.first, a.last, xf);
        } else {  // estimate.rs:152-162, :193-235
            T.linetos += 2ull * a.lineto_lines;
            T.curves += 2ull * a.curve_lines;
            T.curve_count += 2ull * a.curve_count;
            T.segments += 2ull * a.segments;
            for (int end = 0; end < 2; end++) {
                uint32_t cap = end == 0 ? (style_flags & vk::STYLE_FLAGS_START_CAP_MASK) >> 2 : (style_flags & vk::STYLE_FLAGS_END_CAP_MASK);
                if (cap == vk::STYLE_FLAGS_CAP_ROUND) {
                    uint32_t n; double ll;
                    arc_lines(scaled_width, n, ll);
                    T.curves += (uint64_t)a.caps * n;
                    T.curve_count += 1;
                    T.segments += (uint64_t)a.caps * n * segs_for_len(ll);
                } else if (cap == vk::STYLE_FLAGS_CAP_SQUARE) {
                    T.linetos += 3ull * a.caps;
                    T.segments += (uint64_t)segs_for_len(scaled_width) * a.caps + 2ull * segs_for_len(0.5 * scaled_width) * a.caps;
                } else {
                    T.linetos += a.caps;
                    T.segments += (uint64_t)segs_for_len(scaled_width) * a.caps;
                }
            }
            uint32_t join = style_flags & vk::STYLE_FLAGS_JOIN_MASK;
            if (join == vk::STYLE_FLAGS_JOIN_ROUND) {
                uint32_t n; double ll;
                arc_lines(scaled_width, n, ll);
                T.curves += (uint64_t)a.joins * n;
                T.curve_count += 1;
                T.segments += (uint64_t)a.joins * n * segs_for_len(ll);
            } else if (join == vk::STYLE_FLAGS_JOIN_MITER) {
                T.linetos += 2ull * a.joins;
                T.segments += 2ull * a.joins * segs_for_len(scaled_width * 4.0);
            } else {
                T.linetos += a.joins;
                T.segments += (uint64_t)segs_for_len(scaled_width) * a.joins;
            }
            T.linetos += a.joins;  // inner join lines
            T.segments += (uint64_t)segs_for_len(scaled_width) * a.joins;
        }
        // tile rectangle of the path (tile_alloc.wgsl:62-80), from the control points (+ the stroke's reach)
        if (a.x1 >= a.x0 && a.y1 >= a.y0) {
            double r = is_stroke ? 0.5 * scaled_width * 4.0 + 1.0 : 0.0;  // miter limit 4 upper bound on the offset
            double fx0 = std::floor((a.x0 - r) / 16.0), fy0 = std::floor((a.y0 - r) / 16.0);
            double fx1 = std::ceil((a.x1 + r) / 16.0), fy1 = std::ceil((a.y1 + r) / 16.0);
            double cx0 = std::min(std::max(fx0, 0.0), (double)wt), cx1 = std::min(std::max(fx1, 0.0), (double)wt);
            double cy0 = std::min(std::max(fy0, 0.0), (double)ht), cy1 = std::min(std::max(fy1, 0.0), (double)ht);
            uint64_t tiles = (uint64_t)(cx1 - cx0) * (uint64_t)(cy1 - cy0);
            T.tiles += tiles;
            if (tiles) {
                T.bins += (uint64_t)(std::floor((cx1 - 1.0) / 16.0) - std::floor(cx0 / 16.0) + 1.0) *
                          (uint64_t)(std::floor((cy1 - 1.0) / 16.0) - std::floor(cy0 / 16.0) + 1.0);
                uint64_t path_segs = T.segments - seg_before;
                if (!is_stroke) T.ptcl_pairs += tiles;  // tiles of a fill's rectangle
                (void)path_segs;
            }
        }
        a = PathAcc();
    };

    for (size_t i = 0; i < n_tags; i++) {
        const uint32_t t = tags[i];
        if (t == 0u) continue;
        const uint32_t seg = t & vk::PATH_TAG_SEG_TYPE;
        if (seg != 0u) {
            const bool f32 = (t & vk::PATH_TAG_F32) != 0u;
            const size_t pw = f32 ? 2u : 1u;
            V2 p[4];
            bool ok = true;
            for (uint32_t k = 0; k <= seg && ok; k++) ok = read_pt(off + k * pw, f32, p[k]);
            if (ok) {
                a.any = true;
                if (!a.have_first) { a.first = p[0]; a.have_first = true; }
                for (uint32_t k = 0; k <= seg; k++) a.bbox(apply(xf, p[k]));
                const double scale = transform_scale(xf);
                const double offset_fudge = std::max(std::sqrt(is_stroke ? width * scale : 0.0), 1.0);
                if (seg == vk::PATH_TAG_LINETO) {  // estimate.rs:96-101
                    a.joins += 1;
                    a.lineto_lines += 1;
                    a.segments += segs_for_line(p[0], p[1], xf);
                } else if (seg == vk::PATH_TAG_QUADTO) {  // :102-118
                    double lines = offset_fudge * wang_quad(p[0], p[1], p[2], xf);
                    a.curve_lines += sat_u32(std::ceil(lines));
                    a.curve_count += 1;
                    a.joins += 1;
                    double segs = offset_fudge * segs_for_cubic(p[0], lerp(p[1], p[0], 0.333333), lerp(p[1], p[2], 0.333333), p[2], xf);
                    a.segments += sat_u32(std::max(std::ceil(segs), std::ceil(lines)));
                } else {  // :119-135
                    double lines = offset_fudge * wang_cubic(p[0], p[1], p[2], p[3], xf);
                    a.curve_lines += sat_u32(std::ceil(lines));
                    a.curve_count += 1;
                    a.joins += 1;
                    a.segments += sat_u32(std::max(segs_for_cubic(p[0], p[1], p[2], p[3], xf), std::ceil(lines)));
                }
                a.last = p[seg];
                a.have_last = true;
            }
            const uint32_t n_points = seg + ((t & vk::PATH_TAG_SUBPATH_END) ? 1u : 0u);  // path.rs:340-345
            off += (size_t)n_points * pw;
            if (t & vk::PATH_TAG_SUBPATH_END) {  // the next segment opens a new subpath (estimate.rs:76-88 MoveTo)
                a.caps += 1;
                a.joins = a.joins > 0 ? a.joins - 1 : 0;
                a.fill_close_lines += 1;
            }
        }
        if (t & vk::PATH_TAG_TRANSFORM) {
            if (n_xf_seen < n_xf) {
                for (int k = 0; k < 6; k++) {
                    float f;
                    std::memcpy(&f, &w[L.transform_base + n_xf_seen * 6u + k], 4);
                    xf.m[k] = f;
                }
            }
            n_xf_seen++;
        }
        if (t & vk::PATH_TAG_STYLE) {
            if (n_style_seen < n_style) {
                style_flags = w[L.style_base + n_style_seen * 2u];
                float wd;
                std::memcpy(&wd, &w[L.style_base + n_style_seen * 2u + 1u], 4);
                is_stroke = (style_flags & vk::STYLE_FLAGS_STYLE) != 0u;
                width = is_stroke ? std::fabs((double)wd) : 0.0;
            }
            n_style_seen++;
        }
        if (t & vk::PATH_TAG_PATH) finish_path();
    }
    finish_path();

    // tally (estimate.rs:165-190, :252-275), scene-level transform = identity
    const uint64_t curves = std::max<uint64_t>(T.curves, 5u * T.curve_count);
    const uint64_t lines = T.linetos + curves;
    const uint64_t n_segments = std::max(T.segments, lines);
    auto cap32 = [](uint64_t v) { return (uint32_t)std::min<uint64_t>(v, 0xffff0000ull); };
    out->lines = cap32(lines + lines / 8u + 1024u);
    out->seg_counts = cap32(n_segments + n_segments / 8u + 1024u);
    out->segments = out->seg_counts;
    out->tiles = cap32(T.tiles + 1024u);
    out->bin_data = cap32((uint64_t)L.bin_data_start + T.bins + (uint64_t)wb * hb + 1024u);
    // PTCL: not a strict bound (that would be 7 words for every tile of every bounding box) but what scenes need in
    // practice, with robust mode as the net: the fixed blocks, ~2 words per estimated segment (a (path, tile) pair with
    // segments costs 6-7 words and holds 2-4 of them once the estimate's own slack is counted), 7 words for every tile
    // of a FILL's rectangle (interior tiles hold CMD_SOLID + a draw command), and a region's slack per tile
    const uint64_t n_tiles = (uint64_t)wt * ht;
    out->ptcl = cap32(128u * n_tiles + 2u * n_segments + 7u * T.ptcl_pairs + 4096u);
    out->blend_spill = 0u;  // clip depth is not visible in the path streams: left to the caller / auto-grow
    return VELLO_HIP_OK;
}
