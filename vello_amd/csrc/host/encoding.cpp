// See encoding.hpp for the reference map.
#include "encoding.hpp"

#include <cassert>
#include <cmath>
#include <cstring>

namespace vello_encoding {

static inline uint32_t f2bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
static inline float bits2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

Transform Transform::from_kurbo(const kurbo::Affine &a) {
    Transform t;
    for (int i = 0; i < 4; i++) t.matrix[i] = (float)a.c[i];
    t.translation[0] = (float)a.c[4];
    t.translation[1] = (float)a.c[5];
    return t;
}
Transform Transform::operator*(const Transform &o) const {
    Transform r;
    r.matrix[0] = matrix[0] * o.matrix[0] + matrix[2] * o.matrix[1];
    r.matrix[1] = matrix[1] * o.matrix[0] + matrix[3] * o.matrix[1];
    r.matrix[2] = matrix[0] * o.matrix[2] + matrix[2] * o.matrix[3];
    r.matrix[3] = matrix[1] * o.matrix[2] + matrix[3] * o.matrix[3];
    r.translation[0] = matrix[0] * o.translation[0] + matrix[2] * o.translation[1] + translation[0];
    r.translation[1] = matrix[1] * o.translation[0] + matrix[3] * o.translation[1] + translation[1];
    return r;
}
bool Transform::operator==(const Transform &o) const {
    for (int i = 0; i < 4; i++)
        if (matrix[i] != o.matrix[i]) return false;
    return translation[0] == o.translation[0] && translation[1] == o.translation[1];
}

// math.rs:86-119 (Fabian Giesen float_to_half_fast3)
uint16_t f32_to_f16(float val) {
    const uint32_t INF_32 = 255u << 23, INF_16 = 31u << 23, MAGIC = 15u << 23;
    const uint32_t SIGN_MASK = 0x80000000u, ROUND_MASK = ~0xFFFu;
    uint32_t u = f2bits(val);
    uint32_t sign = u & SIGN_MASK;
    u ^= sign;
    uint16_t output;
    if (u >= INF_32) {
        output = (u > INF_32) ? 0x7E00 : 0x7C00;
    } else {
        u &= ROUND_MASK;
        u = f2bits(bits2f(u) * bits2f(MAGIC));
        u -= ROUND_MASK;
        if (u > INF_16) u = INF_16;
        output = (uint16_t)(u >> 13);
    }
    return output | (uint16_t)(sign >> 16);
}
// math.rs:127-150
float f16_to_f32(uint16_t b) {
    uint32_t bits = b;
    const uint32_t MAGIC = 113u << 23, SHIFTED_EXP = 0x7c00u << 13;
    uint32_t o = (bits & 0x7fffu) << 13;
    uint32_t e = SHIFTED_EXP & o;
    o += (127u - 15u) << 23;
    if (e == SHIFTED_EXP) {
        o += (128u - 16u) << 23;
    } else if (e == 0) {
        o += 1u << 23;
        o = f2bits(bits2f(o) - bits2f(MAGIC));
    }
    return bits2f(o | ((bits & 0x8000u) << 16));
}

Style Style::from_fill(Fill fill) { return {fill == Fill::NonZero ? 0u : FLAGS_FILL_BIT, 0.f}; }

std::optional<Style> Style::from_stroke(const kurbo::Stroke &stroke) {
    if (stroke.width == 0.0) return std::nullopt;
    uint32_t style = FLAGS_STYLE_BIT;
    uint32_t join = stroke.join == kurbo::Join::Bevel ? FLAGS_JOIN_BITS_BEVEL
                  : stroke.join == kurbo::Join::Miter ? FLAGS_JOIN_BITS_MITER
                                                      : FLAGS_JOIN_BITS_ROUND;
    auto cap_bits = [](kurbo::Cap c) {
        return c == kurbo::Cap::Butt ? FLAGS_CAP_BITS_BUTT : c == kurbo::Cap::Square ? FLAGS_CAP_BITS_SQUARE : FLAGS_CAP_BITS_ROUND;
    };
    uint32_t start_cap = cap_bits(stroke.start_cap) << 2;
    uint32_t end_cap = cap_bits(stroke.end_cap);
    uint32_t miter_limit = f32_to_f16((float)stroke.miter_limit);
    return Style{style | join | start_cap | end_cap | miter_limit, (float)stroke.width};
}

// color crate: premultiply then to_rgba8 = (x * 255 + 0.5) truncated, saturating
uint32_t Color::premul_rgba8() const {
    auto q = [](float x) -> uint32_t {
        float v = x * 255.0f + 0.5f;
        if (!(v > 0.0f)) return 0u;
        if (v >= 255.0f) return 255u;
        return (uint32_t)v;
    };
    return q(r * a) | (q(g * a) << 8) | (q(b * a) << 16) | (q(a) << 24);
}

// ---------------- PathEncoder (path.rs:426-817) ----------------
static constexpr float EPSILON = 1e-12f;

PathEncoder::PathEncoder(Encoding &enc, bool is_fill) : e_(enc), is_fill_(is_fill) {}

void PathEncoder::move_to(float x, float y) {
    if (is_fill_) close();
    if (state_ == State::MoveTo) {
        e_.path_data.resize(e_.path_data.size() - 2);
    } else if (state_ == State::NonemptySubpath) {
        if (!is_fill_) insert_stroke_cap_marker_segment(false);
        if (!e_.path_tags.empty()) e_.path_tags.back() |= PathTag::SUBPATH_END_BIT;
    }
    first_point_[0] = x;
    first_point_[1] = y;
    e_.path_data.push_back(f2bits(x));
    e_.path_data.push_back(f2bits(y));
    state_ = State::MoveTo;
}

void PathEncoder::line_to(float x, float y) {
    if (state_ == State::Start) {
        if (n_encoded_segments_ == 0) {
            move_to(x, y);
            return;
        }
        move_to(first_point_[0], first_point_[1]);
    }
    if (state_ == State::MoveTo) {
        // start_tangent_for_line (path.rs:779-791)
        float p0x = first_point_[0], p0y = first_point_[1];
        if (std::fabs(x - p0x) > EPSILON || std::fabs(y - p0y) > EPSILON) {
            first_start_tangent_end_[0] = p0x + 1.f / 3.f * (x - p0x);
            first_start_tangent_end_[1] = p0y + 1.f / 3.f * (y - p0y);
        } else {
            return;
        }
    }
    if (is_zero_length_segment(x, y, nullptr, nullptr)) return;
    e_.path_data.push_back(f2bits(x));
    e_.path_data.push_back(f2bits(y));
    e_.path_tags.push_back(PathTag::LINE_TO_F32);
    state_ = State::NonemptySubpath;
    n_encoded_segments_++;
}

void PathEncoder::quad_to(float x1, float y1, float x2, float y2) {
    if (state_ == State::Start) {
        if (n_encoded_segments_ == 0) {
            move_to(x2, y2);
            return;
        }
        move_to(first_point_[0], first_point_[1]);
    }
    if (state_ == State::MoveTo) {
        // start_tangent_for_quad (path.rs:794-812)
        float p0x = first_point_[0], p0y = first_point_[1];
        if (std::fabs(x1 - p0x) > EPSILON || std::fabs(y1 - p0y) > EPSILON) {
            first_start_tangent_end_[0] = x1 + 1.f / 3.f * (p0x - x1);
            first_start_tangent_end_[1] = y1 + 1.f / 3.f * (p0y - y1);
        } else if (std::fabs(x2 - p0x) > EPSILON || std::fabs(y2 - p0y) > EPSILON) {
            first_start_tangent_end_[0] = x1 + 1.f / 3.f * (x2 - x1);
            first_start_tangent_end_[1] = y1 + 1.f / 3.f * (y2 - y1);
        } else {
            return;
        }
    }
    float p2[2] = {x2, y2};
    if (is_zero_length_segment(x1, y1, p2, nullptr)) return;
    for (float v : {x1, y1, x2, y2}) e_.path_data.push_back(f2bits(v));
    e_.path_tags.push_back(PathTag::QUAD_TO_F32);
    state_ = State::NonemptySubpath;
    n_encoded_segments_++;
}

void PathEncoder::cubic_to(float x1, float y1, float x2, float y2, float x3, float y3) {
    if (state_ == State::Start) {
        if (n_encoded_segments_ == 0) {
            move_to(x3, y3);
            return;
        }
        move_to(first_point_[0], first_point_[1]);
    }
    if (state_ == State::MoveTo) {
        // start_tangent_for_curve (path.rs:757-776)
        float p0x = first_point_[0], p0y = first_point_[1];
        if (std::fabs(x1 - p0x) > EPSILON || std::fabs(y1 - p0y) > EPSILON) {
            first_start_tangent_end_[0] = x1;
            first_start_tangent_end_[1] = y1;
        } else if (std::fabs(x2 - p0x) > EPSILON || std::fabs(y2 - p0y) > EPSILON) {
            first_start_tangent_end_[0] = x2;
            first_start_tangent_end_[1] = y2;
        } else if (std::fabs(x3 - p0x) > EPSILON || std::fabs(y3 - p0y) > EPSILON) {
            first_start_tangent_end_[0] = x3;
            first_start_tangent_end_[1] = y3;
        } else {
            return;
        }
    }
    float p2[2] = {x2, y2}, p3[2] = {x3, y3};
    if (is_zero_length_segment(x1, y1, p2, p3)) return;
    for (float v : {x1, y1, x2, y2, x3, y3}) e_.path_data.push_back(f2bits(v));
    e_.path_tags.push_back(PathTag::CUBIC_TO_F32);
    state_ = State::NonemptySubpath;
    n_encoded_segments_++;
}

void PathEncoder::empty_path() {
    for (int i = 0; i < 4; i++) e_.path_data.push_back(f2bits(0.f));
    e_.path_tags.push_back(PathTag::LINE_TO_F32);
    n_encoded_segments_++;
}

void PathEncoder::close() {
    if (state_ == State::Start) return;
    if (state_ == State::MoveTo) {
        e_.path_data.resize(e_.path_data.size() - 2);
        state_ = State::Start;
        return;
    }
    size_t len = e_.path_data.size();
    if (len < 2) return;
    if (e_.path_data[len - 2] != f2bits(first_point_[0]) || e_.path_data[len - 1] != f2bits(first_point_[1])) {
        e_.path_data.push_back(f2bits(first_point_[0]));
        e_.path_data.push_back(f2bits(first_point_[1]));
        e_.path_tags.push_back(PathTag::LINE_TO_F32);
        n_encoded_segments_++;
    }
    if (!is_fill_) insert_stroke_cap_marker_segment(true);
    if (!e_.path_tags.empty()) e_.path_tags.back() |= PathTag::SUBPATH_END_BIT;
    state_ = State::Start;
}

void PathEncoder::path_elements(const kurbo::BezPath &path) {
    for (const auto &el : path.els) {
        switch (el.verb) {
        case kurbo::Verb::MoveTo: move_to((float)el.p[0].x, (float)el.p[0].y); break;
        case kurbo::Verb::LineTo: line_to((float)el.p[0].x, (float)el.p[0].y); break;
        case kurbo::Verb::QuadTo: quad_to((float)el.p[0].x, (float)el.p[0].y, (float)el.p[1].x, (float)el.p[1].y); break;
        case kurbo::Verb::CurveTo:
            cubic_to((float)el.p[0].x, (float)el.p[0].y, (float)el.p[1].x, (float)el.p[1].y, (float)el.p[2].x, (float)el.p[2].y);
            break;
        case kurbo::Verb::ClosePath: close(); break;
        }
    }
}

uint32_t PathEncoder::finish(bool insert_path_marker) {
    if (is_fill_) close();
    if (state_ == State::MoveTo) e_.path_data.resize(e_.path_data.size() - 2);
    if (n_encoded_segments_ != 0) {
        if (!is_fill_ && state_ == State::NonemptySubpath) insert_stroke_cap_marker_segment(false);
        if (!e_.path_tags.empty()) e_.path_tags.back() |= PathTag::SUBPATH_END_BIT;
        e_.n_path_segments += n_encoded_segments_;
        if (insert_path_marker) {
            e_.path_tags.push_back(PathTag::PATH);
            e_.n_paths += 1;
        }
    }
    return n_encoded_segments_;
}

void PathEncoder::insert_stroke_cap_marker_segment(bool is_closed) {
    assert(!is_fill_);
    assert(state_ == State::NonemptySubpath);
    if (is_closed) {
        line_to(first_start_tangent_end_[0], first_start_tangent_end_[1]);
    } else {
        quad_to(first_point_[0], first_point_[1], first_start_tangent_end_[0], first_start_tangent_end_[1]);
    }
}

bool PathEncoder::is_zero_length_segment(float p1x, float p1y, const float *p2, const float *p3) const {
    size_t len = e_.path_data.size();
    float p0x = bits2f(e_.path_data[len - 2]), p0y = bits2f(e_.path_data[len - 1]);
    float p2x = p2 ? p2[0] : p1x, p2y = p2 ? p2[1] : p1y;
    float p3x = p3 ? p3[0] : p1x, p3y = p3 ? p3[1] : p1y;
    float x_min = std::fmin(p0x, std::fmin(p1x, std::fmin(p2x, p3x)));
    float x_max = std::fmax(p0x, std::fmax(p1x, std::fmax(p2x, p3x)));
    float y_min = std::fmin(p0y, std::fmin(p1y, std::fmin(p2y, p3y)));
    float y_max = std::fmax(p0y, std::fmax(p1y, std::fmax(p2y, p3y)));
    return !(x_max - x_min > EPSILON || y_max - y_min > EPSILON);
}

// ---------------- Encoding (encoding.rs) ----------------
void Encoding::reset() {
    path_tags.clear();
    path_data.clear();
    draw_tags.clear();
    draw_data.clear();
    transforms.clear();
    styles.clear();
    resources.reset();
    n_paths = n_path_segments = n_clips = n_open_clips = flags = 0;
}

void Encoding::append(const Encoding &other, const std::optional<Transform> &transform) {
    // encoding.rs:95-152: late-bound resources move with their stream offsets
    {
        const size_t draw_data_base = draw_data.size();
        const size_t stops_base = resources.color_stops.size();
        for (Patch p : other.resources.patches) {
            p.draw_data_offset += draw_data_base;
            if (p.kind == Patch::Kind::Ramp) {
                p.stops_begin += stops_base;
                p.stops_end += stops_base;
            }
            resources.patches.push_back(std::move(p));
        }
        resources.color_stops.insert(resources.color_stops.end(), other.resources.color_stops.begin(), other.resources.color_stops.end());
    }
    path_tags.insert(path_tags.end(), other.path_tags.begin(), other.path_tags.end());
    path_data.insert(path_data.end(), other.path_data.begin(), other.path_data.end());
    draw_tags.insert(draw_tags.end(), other.draw_tags.begin(), other.draw_tags.end());
    draw_data.insert(draw_data.end(), other.draw_data.begin(), other.draw_data.end());
    n_paths += other.n_paths;
    n_path_segments += other.n_path_segments;
    n_clips += other.n_clips;
    n_open_clips += other.n_open_clips;
    flags = other.flags;
    if (transform) {
        for (const auto &x : other.transforms) transforms.push_back(*transform * x);
    } else {
        transforms.insert(transforms.end(), other.transforms.begin(), other.transforms.end());
    }
    styles.insert(styles.end(), other.styles.begin(), other.styles.end());
}

void Encoding::encode_fill_style(Fill fill) { encode_style(Style::from_fill(fill)); }

bool Encoding::encode_stroke_style(const kurbo::Stroke &stroke) {
    auto s = Style::from_stroke(stroke);
    if (!s) return false;
    encode_style(*s);
    return true;
}

void Encoding::encode_style(const Style &style) {
    if ((flags & FORCE_NEXT_STYLE) != 0 || styles.empty() || !(styles.back() == style)) {
        path_tags.push_back(PathTag::STYLE);
        styles.push_back(style);
        flags &= ~FORCE_NEXT_STYLE;
    }
}

bool Encoding::encode_transform(const Transform &t) {
    if ((flags & FORCE_NEXT_TRANSFORM) != 0 || transforms.empty() || !(transforms.back() == t)) {
        path_tags.push_back(PathTag::TRANSFORM);
        transforms.push_back(t);
        flags &= ~FORCE_NEXT_TRANSFORM;
        return true;
    }
    return false;
}

bool Encoding::encode_path_elements(const kurbo::BezPath &path, bool is_fill) {
    PathEncoder enc(*this, is_fill);
    enc.path_elements(path);
    return enc.finish(true) != 0;
}

void Encoding::encode_empty_shape() {
    PathEncoder enc(*this, true);
    enc.empty_path();
    enc.finish(true);
}

void Encoding::encode_color(uint32_t rgba) {
    draw_tags.push_back(DrawTag::COLOR);
    draw_data.push_back(rgba);
}

// ---------------- brushes with late-bound resources (encoding.rs:286-484) ----------------
static void push_words(std::vector<uint32_t> &v, const void *p, size_t n_words) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    v.insert(v.end(), w, w + n_words);
}

Encoding::RampStops Encoding::add_ramp(const std::vector<ColorStop> &stops, float alpha, Extend extend, InterpolationAlphaSpace space,
                                       Color *one) {
    const size_t offset = draw_data.size();
    const size_t stops_start = resources.color_stops.size();
    for (const ColorStop &st : stops) resources.color_stops.push_back(alpha != 1.0f ? st.multiply_alpha(alpha) : st);
    const size_t stops_end = resources.color_stops.size();
    switch (stops_end - stops_start) {
    case 0: return RampStops::Empty;
    case 1:
        *one = resources.color_stops.back().color;
        resources.color_stops.pop_back();
        return RampStops::One;
    default: {
        Patch p;
        p.kind = Patch::Kind::Ramp;
        p.draw_data_offset = offset;
        p.stops_begin = stops_start;
        p.stops_end = stops_end;
        p.extend = extend;
        p.interpolation_alpha_space = space;
        resources.patches.push_back(p);
        return RampStops::Many;
    }
    }
}

void Encoding::encode_linear_gradient(DrawLinearGradient gradient, const std::vector<ColorStop> &stops, float alpha, Extend extend,
                                      InterpolationAlphaSpace space) {
    Color one{};
    switch (add_ramp(stops, alpha, extend, space, &one)) {
    case RampStops::Empty: encode_color(0u); break;  // palette::css::TRANSPARENT
    case RampStops::One: encode_color(one.premul_rgba8()); break;
    case RampStops::Many:
        draw_tags.push_back(DrawTag::LINEAR_GRADIENT);
        push_words(draw_data, &gradient, sizeof gradient / 4);
        break;
    }
}

void Encoding::encode_radial_gradient(DrawRadialGradient gradient, const std::vector<ColorStop> &stops, float alpha, Extend extend,
                                      InterpolationAlphaSpace space) {
    // Skia's epsilon for radii comparison (encoding.rs:396-401)
    const float SKIA_EPSILON = 1.0f / (float)(1 << 12);
    if (gradient.p0[0] == gradient.p1[0] && gradient.p0[1] == gradient.p1[1] && std::fabs(gradient.r0 - gradient.r1) < SKIA_EPSILON) {
        encode_color(0u);
        return;
    }
    Color one{};
    switch (add_ramp(stops, alpha, extend, space, &one)) {
    case RampStops::Empty: encode_color(0u); break;
    case RampStops::One: encode_color(one.premul_rgba8()); break;
    case RampStops::Many:
        draw_tags.push_back(DrawTag::RADIAL_GRADIENT);
        push_words(draw_data, &gradient, sizeof gradient / 4);
        break;
    }
}

void Encoding::encode_sweep_gradient(DrawSweepGradient gradient, const std::vector<ColorStop> &stops, float alpha, Extend extend,
                                     InterpolationAlphaSpace space) {
    const float SKIA_DEGENERATE_THRESHOLD = 1.0f / (float)(1 << 15);
    if (std::fabs(gradient.t0 - gradient.t1) < SKIA_DEGENERATE_THRESHOLD) {
        encode_color(0u);
        return;
    }
    Color one{};
    switch (add_ramp(stops, alpha, extend, space, &one)) {
    case RampStops::Empty: encode_color(0u); break;
    case RampStops::One: encode_color(one.premul_rgba8()); break;
    case RampStops::Many:
        draw_tags.push_back(DrawTag::SWEEP_GRADIENT);
        push_words(draw_data, &gradient, sizeof gradient / 4);
        break;
    }
}

// encode_brush's Gradient arm (encoding.rs:286-345)
void Encoding::encode_gradient(const Gradient &g, float alpha) {
    switch (g.kind) {
    case Gradient::Kind::Linear:
        encode_linear_gradient({0u, {(float)g.p0[0], (float)g.p0[1]}, {(float)g.p1[0], (float)g.p1[1]}}, g.stops, alpha, g.extend,
                               g.interpolation_alpha_space);
        break;
    case Gradient::Kind::Radial:
        encode_radial_gradient({0u, {(float)g.p0[0], (float)g.p0[1]}, {(float)g.p1[0], (float)g.p1[1]}, g.r0, g.r1}, g.stops, alpha,
                               g.extend, g.interpolation_alpha_space);
        break;
    case Gradient::Kind::Sweep: {
        const float TAU = 6.28318530717958647692528676655900577f;
        encode_sweep_gradient({0u, {(float)g.p0[0], (float)g.p0[1]}, g.start_angle / TAU, g.end_angle / TAU}, g.stops, alpha, g.extend,
                              g.interpolation_alpha_space);
        break;
    }
    }
}

void Encoding::encode_image(const ImageBrush &brush, float alpha) {
    const ImageSampler &sm = brush.sampler;
    // (global_alpha * alpha * 255.0).round() as u8: round half away from zero, saturating
    float av = std::round(sm.alpha * alpha * 255.0f);
    uint32_t a8 = !(av > 0.0f) ? 0u : (av >= 255.0f ? 255u : (uint32_t)av);
    Patch p;
    p.kind = Patch::Kind::Image;
    p.draw_data_offset = draw_data.size();
    p.image = brush.image;
    resources.patches.push_back(p);
    draw_tags.push_back(DrawTag::IMAGE);
    draw_data.push_back(0u);  // xy, patched by the Resolver
    draw_data.push_back((brush.image.width << 16) | (brush.image.height & 0xFFFFu));
    draw_data.push_back(((uint32_t)brush.image.format << 15) | ((uint32_t)brush.image.alpha_type << 14) | ((uint32_t)sm.quality << 12) |
                        ((uint32_t)sm.x_extend << 10) | ((uint32_t)sm.y_extend << 8) | a8);
}

void Encoding::encode_blurred_rounded_rect(uint32_t rgba, float width, float height, float radius, float std_dev) {
    draw_tags.push_back(DrawTag::BLUR_RECT);
    draw_data.push_back(rgba);
    push_words(draw_data, &width, 1);
    push_words(draw_data, &height, 1);
    push_words(draw_data, &radius, 1);
    push_words(draw_data, &std_dev, 1);
}

void Encoding::encode_begin_clip(const DrawBeginClip &p) {
    draw_tags.push_back(DrawTag::BEGIN_CLIP);
    draw_data.push_back(p.blend_mode);
    draw_data.push_back(f2bits(p.alpha));
    n_clips += 1;
    n_open_clips += 1;
}

void Encoding::encode_end_clip() {
    if (n_open_clips > 0) {
        draw_tags.push_back(DrawTag::END_CLIP);
        path_tags.push_back(PathTag::PATH);
        n_paths += 1;
        n_clips += 1;
        n_open_clips -= 1;
    }
}

void Encoding::swap_last_path_tags() {
    size_t len = path_tags.size();
    std::swap(path_tags[len - 1], path_tags[len - 2]);
}

// ---------------- resolve (resolve.rs:107-154, :612-640) ----------------
static size_t align_up(size_t len, size_t alignment) { return len + ((0 - len) & (alignment - 1)); }

Layout resolve_solid_paths_only(const Encoding &encoding, std::vector<uint8_t> &data) {
    data.clear();
    Layout layout{};
    layout.n_paths = encoding.n_paths;
    layout.n_clips = encoding.n_clips;
    size_t n_path_tags = encoding.path_tags.size() + encoding.n_open_clips;
    size_t path_tag_padded = align_up(n_path_tags, 4 * 256);
    auto words = [&]() { return (uint32_t)(data.size() / 4); };
    auto push_u32s = [&](const uint32_t *p, size_t n) {
        const uint8_t *b = reinterpret_cast<const uint8_t *>(p);
        data.insert(data.end(), b, b + n * 4);
    };
    layout.path_tag_base = words();
    data.insert(data.end(), encoding.path_tags.begin(), encoding.path_tags.end());
    for (uint32_t i = 0; i < encoding.n_open_clips; i++) data.push_back(PathTag::PATH);
    data.resize(path_tag_padded, 0);
    layout.path_data_base = words();
    push_u32s(encoding.path_data.data(), encoding.path_data.size());
    layout.draw_tag_base = words();
    uint32_t bds = 0;
    for (uint32_t t : encoding.draw_tags) bds += DrawTag::info_size(t);
    layout.bin_data_start = bds;
    push_u32s(encoding.draw_tags.data(), encoding.draw_tags.size());
    for (uint32_t i = 0; i < encoding.n_open_clips; i++) {
        uint32_t t = DrawTag::END_CLIP;
        push_u32s(&t, 1);
    }
    layout.draw_data_base = words();
    push_u32s(encoding.draw_data.data(), encoding.draw_data.size());
    layout.transform_base = words();
    push_u32s(reinterpret_cast<const uint32_t *>(encoding.transforms.data()), encoding.transforms.size() * 6);
    layout.style_base = words();
    push_u32s(reinterpret_cast<const uint32_t *>(encoding.styles.data()), encoding.styles.size() * 2);
    layout.n_draw_objects = layout.n_paths;
    return layout;
}

}  // namespace vello_encoding
