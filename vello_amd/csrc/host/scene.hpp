// Mirror of vello::Scene (vello/src/scene.rs:45-470): the API surface a vello user drives.
// What produces the packed stream for the hot path is restated: fills and GPU strokes with solid, gradient
// and image brushes, blurred rounded rects, clip/blend layers, append.  Glyph runs stay out of scope
// (they reach the GPU as ordinary paths).
#pragma once
#include <optional>

#include "encoding.hpp"

namespace vello {

using kurbo::Affine;
using vello_encoding::Color;
using vello_encoding::Encoding;
using vello_encoding::Fill;

// peniko::BlendMode { mix, compose }
struct BlendMode {
    uint32_t mix = 0;      // Mix::Normal
    uint32_t compose = 3;  // Compose::SrcOver
};

// peniko::BrushRef
struct Brush {
    enum class Kind { Solid, Gradient, Image } kind = Kind::Solid;
    Color color{0.f, 0.f, 0.f, 1.f};
    vello_encoding::Gradient gradient;
    vello_encoding::ImageBrush image;
    Brush() = default;
    Brush(const Color &c) : kind(Kind::Solid), color(c) {}
    Brush(const vello_encoding::Gradient &g) : kind(Kind::Gradient), gradient(g) {}
    Brush(const vello_encoding::ImageBrush &i) : kind(Kind::Image), image(i) {}
};

class Scene {
  public:
    void reset() { encoding_.reset(); }
    const Encoding &encoding() const { return encoding_; }
    Encoding &encoding_mut() { return encoding_; }

    // scene.rs:316-340
    void fill(Fill style, const Affine &transform, const Color &brush, const kurbo::BezPath &shape);
    void fill(Fill style, const Affine &transform, const Brush &brush, const std::optional<Affine> &brush_transform,
              const kurbo::BezPath &shape);
    bool stroke(const kurbo::Stroke &style, const Affine &transform, const Brush &brush, const std::optional<Affine> &brush_transform,
                const kurbo::BezPath &shape);
    // scene.rs:256-309
    void draw_blurred_rounded_rect(const Affine &transform, const kurbo::Rect &rect, const Color &brush, double radius, double std_dev);
    void draw_blurred_rounded_rect_in(const kurbo::BezPath &shape, const Affine &transform, const kurbo::Rect &rect, const Color &brush,
                                      double radius, double std_dev);
    // scene.rs:443-452
    void draw_image(const vello_encoding::ImageBrush &image, const Affine &transform);
    // scene.rs:347-440 (GPU_STROKES = true; dashing is expanded on the CPU by kurbo::dash upstream
    // and is not restated here: a non-empty dash_pattern is rejected)
    bool stroke(const kurbo::Stroke &style, const Affine &transform, const Color &brush, const kurbo::BezPath &shape);
    // scene.rs:105-253
    void push_layer(Fill clip_style, BlendMode blend, float alpha, const Affine &transform, const kurbo::BezPath &clip);
    void push_luminance_mask_layer(Fill clip_style, float alpha, const Affine &transform, const kurbo::BezPath &clip);
    void push_clip_layer(Fill clip_style, const Affine &transform, const kurbo::BezPath &clip);
    // the same three with a stroke as clip style (StyleRef::Stroke, scene.rs:177-187): the layer is clipped to the
    // stroked outline of `clip`; a zero-width stroke suppresses all drawing until the layer is popped.  Returns false
    // for a dashed stroke (kurbo::dash is not restated), in which case nothing was encoded.
    bool push_layer(const kurbo::Stroke &clip_style, BlendMode blend, float alpha, const Affine &transform, const kurbo::BezPath &clip);
    bool push_luminance_mask_layer(const kurbo::Stroke &clip_style, float alpha, const Affine &transform, const kurbo::BezPath &clip);
    bool push_clip_layer(const kurbo::Stroke &clip_style, const Affine &transform, const kurbo::BezPath &clip);
    void pop_layer() { encoding_.encode_end_clip(); }
    // scene.rs:463-469
    void append(const Scene &other, const std::optional<Affine> &transform);

  private:
    void push_layer_inner(const vello_encoding::DrawBeginClip &params, Fill clip_style, const Affine &transform,
                          const kurbo::BezPath &clip);
    bool push_layer_inner(const vello_encoding::DrawBeginClip &params, const kurbo::Stroke &clip_style, const Affine &transform,
                          const kurbo::BezPath &clip);
    void encode_brush(const Brush &brush, float alpha);  // encoding.rs:286-345
    bool stroke_gpu_inner(const kurbo::Stroke &style, const Affine &transform, const kurbo::BezPath &shape);
    Encoding encoding_;
};

}  // namespace vello
