// See renderer.hpp.
#include "renderer.hpp"

#include <cstring>

namespace vello {

Renderer *Renderer::create(const RendererOptions &options, std::string *err) {
    vello_hip_ctx *ctx = nullptr;
    int r = vello_hip_create(options.device, options.antialiasing_support, &options.capacities, &ctx);
    if (r != VELLO_HIP_OK) {
        if (err) *err = vello_hip_last_error(nullptr);
        return nullptr;
    }
    Renderer *re = new Renderer();
    re->ctx_ = ctx;
    return re;
}

Renderer::~Renderer() { vello_hip_destroy(ctx_); }

int Renderer::render_to_texture(const Scene &scene, void *texture, size_t stride, bool is_device, const RenderParams &params) {
    // render::render_full -> Resolver::resolve (vello/src/render.rs:84-112, :165)
    vello_encoding::Layout layout = vello_encoding::resolve_solid_paths_only(scene.encoding(), packed_);
    vello_hip_layout l;
    static_assert(sizeof(l) == sizeof(layout), "Layout");
    std::memcpy(&l, &layout, sizeof l);
    vello_hip_render_params p{params.width, params.height, params.base_color.premul_rgba8(), (uint32_t)params.antialiasing_method};
    int r = vello_hip_render(ctx_, packed_.data(), packed_.size(), &l, &p, nullptr, 0, texture, stride, is_device ? 1 : 0, &bump_);
    if (r != VELLO_HIP_OK) error_ = vello_hip_last_error(ctx_);
    return r;
}

}  // namespace vello
