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
    vello_encoding::Resolved res = resolver_.resolve(scene.encoding(), packed_);
    const vello_encoding::Layout &layout = res.layout;
    // the persistent image atlas (render.rs:160-203)
    if (res.atlas_size) {
        int ar = VELLO_HIP_OK;
        if (res.atlas_resized) ar = vello_hip_resize_image_atlas(ctx_, res.atlas_size, res.atlas_size);
        for (size_t i = 0; ar == VELLO_HIP_OK && res.uploads && i < res.uploads->size(); i++) {
            const vello_encoding::ImageUpload &u = (*res.uploads)[i];
            if (u.image.data && u.image.data->size() >= (size_t)u.image.width * u.image.height * 4u)
                ar = vello_hip_write_image(ctx_, u.x, u.y, u.image.width, u.image.height, u.image.data->data(), 0);
        }
        if (ar != VELLO_HIP_OK) {
            error_ = vello_hip_last_error(ctx_);
            return ar;
        }
    }
    vello_hip_layout l;
    static_assert(sizeof(l) == sizeof(layout), "Layout");
    std::memcpy(&l, &layout, sizeof l);
    vello_hip_render_params p{params.width, params.height, params.base_color.premul_rgba8(), (uint32_t)params.antialiasing_method};
    int r = vello_hip_render(ctx_, packed_.data(), packed_.size(), &l, &p, res.ramps, res.n_ramps, texture, stride, is_device ? 1 : 0, &bump_);
    if (r != VELLO_HIP_OK) error_ = vello_hip_last_error(ctx_);
    return r;
}

}  // namespace vello
