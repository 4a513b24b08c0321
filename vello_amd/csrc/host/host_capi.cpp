// Flat C entry points over the C++ host mirror (kurbo subset, Scene, resolve, Renderer) for the
// Python test / bench harness (vello_amd/*.py binds these with ctypes).  This is harness glue, NOT
// the drop-in boundary: that is include/vello_hip.h.
#include <cstring>
#include <string>

#include "renderer.hpp"

using kurbo::Affine;
using kurbo::BezPath;
using kurbo::Point;
using kurbo::Verb;

namespace {

BezPath path_from_arrays(const uint8_t *verbs, const double *pts, size_t n_verbs) {
    BezPath p;
    size_t k = 0;
    for (size_t i = 0; i < n_verbs; i++) {
        switch ((Verb)verbs[i]) {
        case Verb::MoveTo: p.move_to({pts[k], pts[k + 1]}); k += 2; break;
        case Verb::LineTo: p.line_to({pts[k], pts[k + 1]}); k += 2; break;
        case Verb::QuadTo: p.quad_to({pts[k], pts[k + 1]}, {pts[k + 2], pts[k + 3]}); k += 4; break;
        case Verb::CurveTo: p.curve_to({pts[k], pts[k + 1]}, {pts[k + 2], pts[k + 3]}, {pts[k + 4], pts[k + 5]}); k += 6; break;
        case Verb::ClosePath: p.close_path(); break;
        }
    }
    return p;
}
Affine affine_from(const double *a) {
    if (!a) return Affine::identity();
    return Affine::make(a[0], a[1], a[2], a[3], a[4], a[5]);
}
vello::Color color_from(const float *c) { return vello::Color{c[0], c[1], c[2], c[3]}; }

struct SceneHandle {
    vello::Scene scene;
    std::vector<uint8_t> packed;
};
struct ResolverHandle {
    vello_encoding::Resolver resolver;
    std::vector<uint8_t> packed;
    vello_encoding::Resolved last;
};

}  // namespace

extern "C" {

// ---- BezPath ----
void *vh_bezpath_new() { return new BezPath(); }
void vh_bezpath_free(void *p) { delete (BezPath *)p; }
void *vh_bezpath_from_svg(const char *d) {
    BezPath *p = new BezPath();
    if (!BezPath::from_svg(d, *p)) {
        delete p;
        return nullptr;
    }
    return p;
}
void *vh_bezpath_circle(double cx, double cy, double r, double tol) { return new BezPath(kurbo::path_elements(kurbo::Circle{{cx, cy}, r}, tol)); }
void *vh_bezpath_rect(double x0, double y0, double x1, double y1) { return new BezPath(kurbo::path_elements(kurbo::Rect{x0, y0, x1, y1}, 0.1)); }
void *vh_bezpath_rounded_rect(double x0, double y0, double x1, double y1, double radius, double tol) {
    return new BezPath(kurbo::path_elements(kurbo::RoundedRect{{x0, y0, x1, y1}, radius}, tol));
}
void *vh_bezpath_line(double x0, double y0, double x1, double y1) { return new BezPath(kurbo::path_elements(kurbo::Line{{x0, y0}, {x1, y1}}, 0.1)); }
size_t vh_bezpath_n_verbs(void *p) { return ((BezPath *)p)->els.size(); }
size_t vh_bezpath_n_coords(void *p) {
    size_t n = 0;
    for (auto &e : ((BezPath *)p)->els) n += e.verb == Verb::ClosePath ? 0 : e.verb == Verb::QuadTo ? 4 : e.verb == Verb::CurveTo ? 6 : 2;
    return n;
}
void vh_bezpath_copy(void *p, uint8_t *verbs, double *pts) {
    size_t k = 0, i = 0;
    for (auto &e : ((BezPath *)p)->els) {
        verbs[i++] = (uint8_t)e.verb;
        int np = e.verb == Verb::ClosePath ? 0 : e.verb == Verb::QuadTo ? 2 : e.verb == Verb::CurveTo ? 3 : 1;
        for (int j = 0; j < np; j++) {
            pts[k++] = e.p[j].x;
            pts[k++] = e.p[j].y;
        }
    }
}

// ---- Scene (vello/src/scene.rs) ----
void *vh_scene_new() { return new SceneHandle(); }
void vh_scene_free(void *s) { delete (SceneHandle *)s; }
void vh_scene_reset(void *s) { ((SceneHandle *)s)->scene.reset(); }
void vh_scene_fill(void *s, int fill_rule, const double *affine, const float *color, const uint8_t *verbs, const double *pts, size_t n) {
    ((SceneHandle *)s)->scene.fill((vello::Fill)fill_rule, affine_from(affine), color_from(color), path_from_arrays(verbs, pts, n));
}
int vh_scene_stroke(void *s, double width, int join, double miter_limit, int start_cap, int end_cap, const double *affine,
                    const float *color, const uint8_t *verbs, const double *pts, size_t n) {
    kurbo::Stroke st;
    st.width = width;
    st.join = (kurbo::Join)join;
    st.miter_limit = miter_limit;
    st.start_cap = (kurbo::Cap)start_cap;
    st.end_cap = (kurbo::Cap)end_cap;
    return ((SceneHandle *)s)->scene.stroke(st, affine_from(affine), color_from(color), path_from_arrays(verbs, pts, n)) ? 0 : -1;
}
void vh_scene_push_layer(void *s, int fill_rule, uint32_t mix, uint32_t compose, float alpha, const double *affine, const uint8_t *verbs,
                         const double *pts, size_t n) {
    ((SceneHandle *)s)->scene.push_layer((vello::Fill)fill_rule, vello::BlendMode{mix, compose}, alpha, affine_from(affine),
                                         path_from_arrays(verbs, pts, n));
}
void vh_scene_push_luminance_mask_layer(void *s, int fill_rule, float alpha, const double *affine, const uint8_t *verbs, const double *pts,
                                        size_t n) {
    ((SceneHandle *)s)->scene.push_luminance_mask_layer((vello::Fill)fill_rule, alpha, affine_from(affine), path_from_arrays(verbs, pts, n));
}
void vh_scene_push_clip_layer(void *s, int fill_rule, const double *affine, const uint8_t *verbs, const double *pts, size_t n) {
    ((SceneHandle *)s)->scene.push_clip_layer((vello::Fill)fill_rule, affine_from(affine), path_from_arrays(verbs, pts, n));
}
// layers clipped to a stroked outline (scene.rs:177-187).  kind: 0 push_layer (mix, compose, alpha), 1 luminance mask
// (alpha), 2 clip layer.  Returns -1 for a dashed stroke (nothing encoded).
int vh_scene_push_layer_stroked(void *s, int kind, double width, int join, double miter_limit, int start_cap, int end_cap, uint32_t mix,
                                uint32_t compose, float alpha, const double *affine, const uint8_t *verbs, const double *pts, size_t n) {
    kurbo::Stroke st;
    st.width = width;
    st.join = (kurbo::Join)join;
    st.miter_limit = miter_limit;
    st.start_cap = (kurbo::Cap)start_cap;
    st.end_cap = (kurbo::Cap)end_cap;
    vello::Scene &scene = ((SceneHandle *)s)->scene;
    const kurbo::BezPath path = path_from_arrays(verbs, pts, n);
    bool ok;
    if (kind == 0) ok = scene.push_layer(st, vello::BlendMode{mix, compose}, alpha, affine_from(affine), path);
    else if (kind == 1) ok = scene.push_luminance_mask_layer(st, alpha, affine_from(affine), path);
    else ok = scene.push_clip_layer(st, affine_from(affine), path);
    return ok ? 0 : -1;
}
void vh_scene_pop_layer(void *s) { ((SceneHandle *)s)->scene.pop_layer(); }
void vh_scene_append(void *s, void *other, const double *affine) {
    std::optional<Affine> t;
    if (affine) t = affine_from(affine);
    ((SceneHandle *)s)->scene.append(((SceneHandle *)other)->scene, t);
}

// ---- brushes (peniko::BrushRef) ----
void *vh_brush_solid(const float *color) { return new vello::Brush(color_from(color)); }
// kind 0 linear (p = x0 y0 x1 y1), 1 radial (p = cx0 cy0 cx1 cy1 r0 r1), 2 sweep (p = cx cy start_angle end_angle);
// stops = n_stops x (offset, r, g, b, a)
void *vh_brush_gradient(int kind, const double *p, uint32_t extend, uint32_t alpha_space, const float *stops, size_t n_stops) {
    vello_encoding::Gradient g;
    g.kind = (vello_encoding::Gradient::Kind)kind;
    g.p0[0] = p[0];
    g.p0[1] = p[1];
    if (kind == 0 || kind == 1) {
        g.p1[0] = p[2];
        g.p1[1] = p[3];
    }
    if (kind == 1) {
        g.r0 = (float)p[4];
        g.r1 = (float)p[5];
    }
    if (kind == 2) {
        g.start_angle = (float)p[2];
        g.end_angle = (float)p[3];
    }
    g.extend = (vello_encoding::Extend)extend;
    g.interpolation_alpha_space = (vello_encoding::InterpolationAlphaSpace)alpha_space;
    for (size_t i = 0; i < n_stops; i++)
        g.stops.push_back({stops[5 * i], vello::Color{stops[5 * i + 1], stops[5 * i + 2], stops[5 * i + 3], stops[5 * i + 4]}});
    return new vello::Brush(g);
}
void *vh_brush_image(uint64_t id, uint32_t width, uint32_t height, uint32_t format, uint32_t alpha_type, const uint8_t *rgba8,
                     uint32_t x_extend, uint32_t y_extend, uint32_t quality, float alpha) {
    vello_encoding::ImageBrush b;
    b.image.id = id;
    b.image.width = width;
    b.image.height = height;
    b.image.format = (vello_encoding::ImageFormat)format;
    b.image.alpha_type = (vello_encoding::ImageAlphaType)alpha_type;
    b.image.data = std::make_shared<const std::vector<uint8_t>>(rgba8, rgba8 + (size_t)width * height * 4u);
    b.sampler.x_extend = (vello_encoding::Extend)x_extend;
    b.sampler.y_extend = (vello_encoding::Extend)y_extend;
    b.sampler.quality = (vello_encoding::ImageQuality)quality;
    b.sampler.alpha = alpha;
    return new vello::Brush(b);
}
void vh_brush_free(void *b) { delete (vello::Brush *)b; }
void vh_scene_fill_brush(void *s, int fill_rule, const double *affine, void *brush, const double *brush_affine, const uint8_t *verbs,
                         const double *pts, size_t n) {
    std::optional<Affine> bt;
    if (brush_affine) bt = affine_from(brush_affine);
    ((SceneHandle *)s)->scene.fill((vello::Fill)fill_rule, affine_from(affine), *(vello::Brush *)brush, bt, path_from_arrays(verbs, pts, n));
}
int vh_scene_stroke_brush(void *s, double width, int join, double miter_limit, int start_cap, int end_cap, const double *affine,
                          void *brush, const double *brush_affine, const uint8_t *verbs, const double *pts, size_t n) {
    kurbo::Stroke st;
    st.width = width;
    st.join = (kurbo::Join)join;
    st.miter_limit = miter_limit;
    st.start_cap = (kurbo::Cap)start_cap;
    st.end_cap = (kurbo::Cap)end_cap;
    std::optional<Affine> bt;
    if (brush_affine) bt = affine_from(brush_affine);
    return ((SceneHandle *)s)->scene.stroke(st, affine_from(affine), *(vello::Brush *)brush, bt, path_from_arrays(verbs, pts, n)) ? 0 : -1;
}
void vh_scene_draw_blurred_rounded_rect(void *s, const double *affine, const double *rect, const float *color, double radius,
                                        double std_dev) {
    ((SceneHandle *)s)->scene.draw_blurred_rounded_rect(affine_from(affine), kurbo::Rect{rect[0], rect[1], rect[2], rect[3]},
                                                        color_from(color), radius, std_dev);
}
void vh_scene_draw_blurred_rounded_rect_in(void *s, const uint8_t *verbs, const double *pts, size_t n, const double *affine,
                                           const double *rect, const float *color, double radius, double std_dev) {
    ((SceneHandle *)s)->scene.draw_blurred_rounded_rect_in(path_from_arrays(verbs, pts, n), affine_from(affine),
                                                           kurbo::Rect{rect[0], rect[1], rect[2], rect[3]}, color_from(color), radius,
                                                           std_dev);
}
void vh_scene_draw_image(void *s, void *brush, const double *affine) {
    ((SceneHandle *)s)->scene.draw_image(((vello::Brush *)brush)->image, affine_from(affine));
}
size_t vh_scene_n_patches(void *s) { return ((SceneHandle *)s)->scene.encoding().resources.patches.size(); }

// ---- Resolver (resolve.rs:172-393): ramps + image atlas placement + packed scene ----
void *vh_resolver_new() { return new ResolverHandle(); }
// image_cache.rs:65 new_with_sizes: a resolver whose atlas starts / stops growing at the given sides
void *vh_resolver_new_with_atlas_sizes(uint32_t initial_size, uint32_t max_size) {
    ResolverHandle *h = new ResolverHandle();
    h->resolver = vello_encoding::Resolver(initial_size, max_size);
    return h;
}
void vh_resolver_free(void *r) { delete (ResolverHandle *)r; }
// info_out: n_ramps, atlas_size, atlas_resized, n_uploads, evicted.  Pointers stay valid until the next resolve on this
// resolver.
size_t vh_resolver_resolve(void *r, void *scene, const uint8_t **packed, uint32_t layout_out[10], const uint32_t **ramps,
                           uint32_t info_out[5]) {
    ResolverHandle *h = (ResolverHandle *)r;
    h->last = h->resolver.resolve(((SceneHandle *)scene)->scene.encoding(), h->packed);
    std::memcpy(layout_out, &h->last.layout, sizeof h->last.layout);
    *packed = h->packed.data();
    *ramps = h->last.ramps;
    info_out[0] = h->last.n_ramps;
    info_out[1] = h->last.atlas_size;
    info_out[2] = h->last.atlas_resized ? 1u : 0u;
    info_out[3] = h->last.uploads ? (uint32_t)h->last.uploads->size() : 0u;
    info_out[4] = h->last.evicted;
    return h->packed.size();
}
// Resolver::mark_image_dirty (resolve.rs:173-179): the blob with this id changed; re-upload it when next used
void vh_resolver_mark_image_dirty(void *r, uint64_t image_id) {
    vello_encoding::ImageData image;
    image.id = image_id;
    ((ResolverHandle *)r)->resolver.mark_image_dirty(image);
}
// n_resident, atlas size
void vh_resolver_image_cache_info(void *r, uint32_t out[2]) {
    const vello_encoding::ImageCache &c = ((ResolverHandle *)r)->resolver.image_cache();
    out[0] = (uint32_t)c.n_resident();
    out[1] = c.size();
}
const uint8_t *vh_resolver_upload(void *r, uint32_t i, uint32_t xywh_out[4]) {
    ResolverHandle *h = (ResolverHandle *)r;
    const vello_encoding::ImageUpload &u = (*h->last.uploads)[i];
    xywh_out[0] = u.x;
    xywh_out[1] = u.y;
    xywh_out[2] = u.image.width;
    xywh_out[3] = u.image.height;
    return u.image.data ? u.image.data->data() : nullptr;
}

// stream access: 0 path_tags(u8) 1 path_data(u32) 2 draw_tags(u32) 3 draw_data(u32) 4 transforms(6 f32) 5 styles(2 u32)
size_t vh_scene_stream_bytes(void *s, int which) {
    const auto &e = ((SceneHandle *)s)->scene.encoding();
    switch (which) {
    case 0: return e.path_tags.size();
    case 1: return e.path_data.size() * 4;
    case 2: return e.draw_tags.size() * 4;
    case 3: return e.draw_data.size() * 4;
    case 4: return e.transforms.size() * 24;
    case 5: return e.styles.size() * 8;
    }
    return 0;
}
void vh_scene_stream_copy(void *s, int which, void *dst) {
    const auto &e = ((SceneHandle *)s)->scene.encoding();
    const void *src = nullptr;
    switch (which) {
    case 0: src = e.path_tags.data(); break;
    case 1: src = e.path_data.data(); break;
    case 2: src = e.draw_tags.data(); break;
    case 3: src = e.draw_data.data(); break;
    case 4: src = e.transforms.data(); break;
    case 5: src = e.styles.data(); break;
    }
    size_t n = vh_scene_stream_bytes(s, which);
    if (src && n) std::memcpy(dst, src, n);
}
void vh_scene_counts(void *s, uint32_t out[4]) {
    const auto &e = ((SceneHandle *)s)->scene.encoding();
    out[0] = e.n_paths; out[1] = e.n_path_segments; out[2] = e.n_clips; out[3] = e.n_open_clips;
}

// Resolver::resolve for encodings without late-bound resources (resolve.rs:107-154).
// Returns the packed size; the bytes stay owned by the scene handle until the next resolve.
size_t vh_scene_resolve(void *s, const uint8_t **packed, uint32_t layout_out[10]) {
    SceneHandle *h = (SceneHandle *)s;
    vello_encoding::Layout l = vello_encoding::resolve_solid_paths_only(h->scene.encoding(), h->packed);
    std::memcpy(layout_out, &l, sizeof l);
    *packed = h->packed.data();
    return h->packed.size();
}

// ---- encoding known-answer helpers (math.rs:152-280, draw.rs:281-297, path.rs:847-877) ----
uint16_t vh_f32_to_f16(float v) { return vello_encoding::f32_to_f16(v); }
float vh_f16_to_f32(uint16_t v) { return vello_encoding::f16_to_f32(v); }
uint32_t vh_color_premul_rgba8(const float *c) { return color_from(c).premul_rgba8(); }
void vh_style_from_stroke(double width, int join, double miter_limit, int start_cap, int end_cap, uint32_t out[2]) {
    kurbo::Stroke st;
    st.width = width; st.join = (kurbo::Join)join; st.miter_limit = miter_limit;
    st.start_cap = (kurbo::Cap)start_cap; st.end_cap = (kurbo::Cap)end_cap;
    auto s = vello_encoding::Style::from_stroke(st);
    out[0] = s ? s->flags_and_miter_limit : 0;
    float w = s ? s->line_width : 0.f;
    std::memcpy(&out[1], &w, 4);
}

// ---- Renderer (vello/src/lib.rs:432-515) ----
void *vh_renderer_new(int device, uint32_t aa_mask, const vello_hip_capacities *caps, char *err, size_t err_len) {
    vello::RendererOptions o;
    o.device = device;
    o.antialiasing_support = aa_mask;
    if (caps) o.capacities = *caps;
    std::string e;
    vello::Renderer *r = vello::Renderer::create(o, &e);
    if (!r && err && err_len) {
        std::strncpy(err, e.c_str(), err_len - 1);
        err[err_len - 1] = 0;
    }
    return r;
}
void vh_renderer_free(void *r) { delete (vello::Renderer *)r; }
int vh_renderer_render_to_texture(void *r, void *scene, void *texture, size_t stride, int is_device, uint32_t width, uint32_t height,
                                  const float *base_color, uint32_t aa) {
    vello::RenderParams p;
    p.base_color = color_from(base_color);
    p.width = width;
    p.height = height;
    p.antialiasing_method = (vello::AaConfig)aa;
    return ((vello::Renderer *)r)->render_to_texture(((SceneHandle *)scene)->scene, texture, stride, is_device != 0, p);
}
const char *vh_renderer_error(void *r) { return ((vello::Renderer *)r)->error().c_str(); }
void *vh_renderer_engine(void *r) { return ((vello::Renderer *)r)->engine(); }
void vh_renderer_last_bump(void *r, vello_hip_bump *out) { *out = ((vello::Renderer *)r)->last_bump(); }

}  // extern "C"
