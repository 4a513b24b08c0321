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

void Scene::append(const Scene &other, const std::optional<Affine> &transform) {
    std::optional<Transform> t;
    if (transform) t = Transform::from_kurbo(*transform);
    encoding_.append(other.encoding_, t);
}

}  // namespace vello
