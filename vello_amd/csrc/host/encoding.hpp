// Host-side mirror of vello_encoding (the input contract of the hot path).
// In the real drop-in this layer stays Rust; this C++ restatement exists because the image has
// no Rust toolchain, and it keeps the reference's names, argument meaning and stream layout:
//   Encoding       vello_encoding/src/encoding.rs:26-53
//   PathEncoder    vello_encoding/src/path.rs:426-817
//   Style          vello_encoding/src/path.rs:13-118
//   Transform      vello_encoding/src/math.rs:12-74
//   DrawTag        vello_encoding/src/draw.rs:15-51
//   Layout/resolve vello_encoding/src/resolve.rs:18-154
//   RenderConfig   vello_encoding/src/config.rs:124-273
#pragma once
#include <cstdint>
#include <optional>
#include <vector>

#include "kurbo.hpp"

namespace vello_encoding {

struct Transform {
    float matrix[4];
    float translation[2];
    static Transform identity() { return {{1.f, 0.f, 0.f, 1.f}, {0.f, 0.f}}; }
    static Transform from_kurbo(const kurbo::Affine &a);
    Transform operator*(const Transform &o) const;  // math.rs:50-74
    bool operator==(const Transform &o) const;
};

enum class Fill : uint8_t { NonZero = 0, EvenOdd = 1 };

struct Style {
    uint32_t flags_and_miter_limit;
    float line_width;
    static constexpr uint32_t FLAGS_STYLE_BIT = 0x80000000u;
    static constexpr uint32_t FLAGS_FILL_BIT = 0x40000000u;
    static constexpr uint32_t FLAGS_JOIN_BITS_BEVEL = 0u;
    static constexpr uint32_t FLAGS_JOIN_BITS_MITER = 0x10000000u;
    static constexpr uint32_t FLAGS_JOIN_BITS_ROUND = 0x20000000u;
    static constexpr uint32_t FLAGS_CAP_BITS_BUTT = 0u;
    static constexpr uint32_t FLAGS_CAP_BITS_SQUARE = 0x01000000u;
    static constexpr uint32_t FLAGS_CAP_BITS_ROUND = 0x02000000u;
    static Style from_fill(Fill fill);
    static std::optional<Style> from_stroke(const kurbo::Stroke &stroke);
    bool operator==(const Style &o) const { return flags_and_miter_limit == o.flags_and_miter_limit && line_width == o.line_width; }
};

uint16_t f32_to_f16(float val);  // math.rs:86-119
float f16_to_f32(uint16_t bits); // math.rs:127-150

namespace PathTag {
constexpr uint8_t LINE_TO_F32 = 0x9, QUAD_TO_F32 = 0xa, CUBIC_TO_F32 = 0xb;
constexpr uint8_t TRANSFORM = 0x20, PATH = 0x10, STYLE = 0x40, SUBPATH_END_BIT = 0x4;
}  // namespace PathTag

namespace DrawTag {
constexpr uint32_t NOP = 0, COLOR = 0x44, LINEAR_GRADIENT = 0x114, RADIAL_GRADIENT = 0x29c, SWEEP_GRADIENT = 0x254;
constexpr uint32_t IMAGE = 0x28C, BLUR_RECT = 0x2d4, BEGIN_CLIP = 0x49, END_CLIP = 0x21;
inline uint32_t info_size(uint32_t tag) { return (tag >> 6) & 0xf; }
}  // namespace DrawTag

// peniko::Color (sRGB, straight alpha, f32 components)
struct Color {
    float r, g, b, a;
    static Color from_rgba8(uint8_t r, uint8_t g, uint8_t b, uint8_t a) { return {r / 255.f, g / 255.f, b / 255.f, a / 255.f}; }
    static Color from_rgb8(uint8_t r, uint8_t g, uint8_t b) { return from_rgba8(r, g, b, 255); }
    Color multiply_alpha(float alpha) const { return {r, g, b, a * alpha}; }
    // color.premultiply().to_rgba8().to_u32(): R in the low byte (draw.rs:70-82)
    uint32_t premul_rgba8() const;
};

// peniko::BlendMode -> DrawBeginClip (draw.rs:191-236)
struct DrawBeginClip {
    uint32_t blend_mode;
    float alpha;
    static constexpr uint32_t LUMINANCE_MASK_BLEND_MODE = 0x10000u;
    static constexpr uint32_t CLIP_BLEND_MODE = 0x8003u;
    static DrawBeginClip make(uint32_t mix, uint32_t compose, float alpha) { return {(mix << 8) | compose, alpha}; }
    static DrawBeginClip luminance_mask(float alpha) { return {LUMINANCE_MASK_BLEND_MODE, alpha}; }
    static DrawBeginClip clip() { return {CLIP_BLEND_MODE, 1.0f}; }
};

struct Encoding;

class PathEncoder {
  public:
    PathEncoder(Encoding &enc, bool is_fill);
    void move_to(float x, float y);
    void line_to(float x, float y);
    void quad_to(float x1, float y1, float x2, float y2);
    void cubic_to(float x1, float y1, float x2, float y2, float x3, float y3);
    void empty_path();
    void close();
    void path_elements(const kurbo::BezPath &path);
    uint32_t finish(bool insert_path_marker);

  private:
    enum class State { Start, MoveTo, NonemptySubpath };
    void insert_stroke_cap_marker_segment(bool is_closed);
    bool is_zero_length_segment(float p1x, float p1y, const float *p2, const float *p3) const;
    Encoding &e_;
    float first_point_[2] = {0, 0};
    float first_start_tangent_end_[2] = {0, 0};
    State state_ = State::Start;
    uint32_t n_encoded_segments_ = 0;
    bool is_fill_;
};

struct Encoding {
    std::vector<uint8_t> path_tags;
    std::vector<uint32_t> path_data;
    std::vector<uint32_t> draw_tags;
    std::vector<uint32_t> draw_data;
    std::vector<Transform> transforms;
    std::vector<Style> styles;
    uint32_t n_paths = 0, n_path_segments = 0, n_clips = 0, n_open_clips = 0, flags = 0;
    static constexpr uint32_t FORCE_NEXT_TRANSFORM = 1, FORCE_NEXT_STYLE = 2;

    bool is_empty() const { return path_tags.empty(); }
    void reset();
    void append(const Encoding &other, const std::optional<Transform> &transform);
    void encode_fill_style(Fill fill);
    bool encode_stroke_style(const kurbo::Stroke &stroke);
    bool encode_transform(const Transform &t);
    bool encode_path_elements(const kurbo::BezPath &path, bool is_fill);
    void encode_empty_shape();
    void encode_color(uint32_t premul_rgba8);
    void encode_begin_clip(const DrawBeginClip &p);
    void encode_end_clip();
    void swap_last_path_tags();

  private:
    void encode_style(const Style &s);
};

struct Layout {
    uint32_t n_draw_objects, n_paths, n_clips, bin_data_start;
    uint32_t path_tag_base, path_data_base, draw_tag_base, draw_data_base, transform_base, style_base;
    uint32_t path_tags_size() const { return (path_data_base - path_tag_base) * 4u; }
};
static_assert(sizeof(Layout) == 40, "Layout");

// resolve.rs:107-154
Layout resolve_solid_paths_only(const Encoding &encoding, std::vector<uint8_t> &packed);

}  // namespace vello_encoding
