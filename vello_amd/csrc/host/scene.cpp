// See scene.hpp.
#include "scene.hpp"

#include <algorithm>

namespace vello {

using vello_encoding::DrawBeginClip;
using vello_encoding::Transform;

void Scene::fill(Fill style, const Affine &transform, const Color &brush, const kurbo::BezPath &shape) {
    Transform t = Transform::from_kurbo(transform);
    encoding_.encode_transform(t);
    encoding_.encode_fill_style(style);
    if (encoding_.encode_path_elements(shape, true)) {
        encoding_.encode_color(brush.premul_rgba8());
    }
}

void Scene::encode_brush(const Brush &brush, float alpha) {
    switch (brush.kind) {
    case Brush::Kind::Solid: {
        Color c = alpha != 1.0f ? brush.color.multiply_alpha(alpha) : brush.color;
        encoding_.encode_color(c.premul_rgba8());
        break;
    }
    case Brush::Kind::Gradient: encoding_.encode_gradient(brush.gradient, alpha); break;
    case Brush::Kind::Image: encoding_.encode_image(brush.image, alpha); break;
    }
}

void Scene::fill(Fill style, const Affine &transform, const Brush &brush, const std::optional<Affine> &brush_transform,
                 const kurbo::BezPath &shape) {
    Transform t = Transform::from_kurbo(transform);
    encoding_.encode_transform(t);
    encoding_.encode_fill_style(style);
    if (encoding_.encode_path_elements(shape, true)) {
        if (brush_transform && encoding_.encode_transform(Transform::from_kurbo(transform * *brush_transform))) {
            encoding_.swap_last_path_tags();
        }
        encode_brush(brush, 1.0f);
    }
}

bool Scene::stroke(const kurbo::Stroke &style, const Affine &transform, const Brush &brush, const std::optional<Affine> &brush_transform,
                   const kurbo::BezPath &shape) {
    if (style.width == 0.) return true;
    if (!style.dash_pattern.empty()) return false;
    if (stroke_gpu_inner(style, transform, shape)) {
        if (brush_transform && encoding_.encode_transform(Transform::from_kurbo(transform * *brush_transform))) {
            encoding_.swap_last_path_tags();
        }
        encode_brush(brush, 1.0f);
    }
    return true;
}

void Scene::draw_blurred_rounded_rect(const Affine &transform, const kurbo::Rect &rect, const Color &brush, double radius,
                                      double std_dev) {
    // the gaussian's support is cut off at 2.5 sigma (scene.rs:264-269)
    double k = 2.5 * std_dev;
    kurbo::Rect shape{rect.x0 - k, rect.y0 - k, rect.x1 + k, rect.y1 + k};
    draw_blurred_rounded_rect_in(kurbo::path_elements(shape, 0.1), transform, rect, brush, radius, std_dev);
}

void Scene::draw_blurred_rounded_rect_in(const kurbo::BezPath &shape, const Affine &transform, const kurbo::Rect &rect,
                                         const Color &brush, double radius, double std_dev) {
    Transform t = Transform::from_kurbo(transform);
    encoding_.encode_transform(t);
    encoding_.encode_fill_style(Fill::NonZero);
    if (encoding_.encode_path_elements(shape, true)) {
        Affine center = Affine::translate(0.5 * (rect.x0 + rect.x1), 0.5 * (rect.y0 + rect.y1));
        if (encoding_.encode_transform(Transform::from_kurbo(transform * center))) encoding_.swap_last_path_tags();
        encoding_.encode_blurred_rounded_rect(brush.premul_rgba8(), (float)(rect.x1 - rect.x0), (float)(rect.y1 - rect.y0), (float)radius,
                                              (float)std_dev);
    }
}

void Scene::draw_image(const vello_encoding::ImageBrush &image, const Affine &transform) {
    kurbo::Rect rect{0.0, 0.0, (double)image.image.width, (double)image.image.height};
    fill(Fill::NonZero, transform, Brush(image), std::nullopt, kurbo::path_elements(rect, 0.1));
}

bool Scene::stroke_gpu_inner(const kurbo::Stroke &style, const Affine &transform, const kurbo::BezPath &shape) {
    Transform t = Transform::from_kurbo(transform);
    encoding_.encode_transform(t);
    encoding_.encode_stroke_style(style);
    return encoding_.encode_path_elements(shape, false);
}

bool Scene::stroke(const kurbo::Stroke &style, const Affine &transform, const Color &brush, const kurbo::BezPath &shape) {
    if (style.width == 0.) return true;
    if (!style.dash_pattern.empty()) return false;
    if (stroke_gpu_inner(style, transform, shape)) {
        encoding_.encode_color(brush.premul_rgba8());
    }
    return true;
}

void Scene::push_layer_inner(const DrawBeginClip &params, Fill clip_style, const Affine &transform, const kurbo::BezPath &clip) {
    Transform t = Transform::from_kurbo(transform);
    encoding_.encode_transform(t);
    encoding_.encode_fill_style(clip_style);
    bool encoded = encoding_.encode_path_elements(clip, true);
    if (!encoded) encoding_.encode_empty_shape();
    encoding_.encode_begin_clip(params);
}

void Scene::push_layer(Fill clip_style, BlendMode blend, float alpha, const Affine &transform, const kurbo::BezPath &clip) {
    push_layer_inner(DrawBeginClip::make(blend.mix, blend.compose, std::clamp(alpha, 0.0f, 1.0f)), clip_style, transform, clip);
}

void Scene::push_luminance_mask_layer(Fill clip_style, float alpha, const Affine &transform, const kurbo::BezPath &clip) {
    push_layer_inner(DrawBeginClip::luminance_mask(std::clamp(alpha, 0.0f, 1.0f)), clip_style, transform, clip);
}

void Scene::push_clip_layer(Fill clip_style, const Affine &transform, const kurbo::BezPath &clip) {
    push_layer_inner(DrawBeginClip::clip(), clip_style, transform, clip);
}

// scene.rs:159-215, StyleRef::Stroke arm
bool Scene::push_layer_inner(const DrawBeginClip &params, const kurbo::Stroke &clip_style, const Affine &transform,
                             const kurbo::BezPath &clip) {
    if (!clip_style.dash_pattern.empty()) return false;
    bool encoded;
    if (clip_style.width == 0.) {
        encoding_.encode_fill_style(Fill::NonZero);
        encoded = false;
    } else {
        encoded = stroke_gpu_inner(clip_style, transform, clip);
    }
    if (!encoded) encoding_.encode_empty_shape();
    encoding_.encode_begin_clip(params);
    return true;
}

bool Scene::push_layer(const kurbo::Stroke &clip_style, BlendMode blend, float alpha, const Affine &transform, const kurbo::BezPath &clip) {
    return push_layer_inner(DrawBeginClip::make(blend.mix, blend.compose, std::clamp(alpha, 0.0f, 1.0f)), clip_style, transform, clip);
}

bool Scene::push_luminance_mask_layer(const kurbo::Stroke &clip_style, float alpha, const Affine &transform, const kurbo::BezPath &clip) {
    return push_layer_inner(DrawBeginClip::luminance_mask(std::clamp(alpha, 0.0f, 1.0f)), clip_style, transform, clip);
}

bool Scene::push_clip_layer(const kurbo::Stroke &clip_style, const Affine &transform, const kurbo::BezPath &clip) {
    return push_layer_inner(DrawBeginClip::clip(), clip_style, transform, clip);
}

void Scene::append(const Scene &other, const std::optional<Affine> &transform) {
    std::optional<Transform> t;
    if (transform) t = Transform::from_kurbo(*transform);
    encoding_.append(other.encoding_, t);
}

}  // namespace vello
