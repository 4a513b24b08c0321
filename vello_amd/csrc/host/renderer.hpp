// Mirror of vello::Renderer / RenderParams / AaConfig (vello/src/lib.rs:175-193, :357-369, :432-515).
// render_to_texture = Resolver::resolve (host) + the C-ABI call that replaces
// WgpuEngine::run_recording.  In the real drop-in this class is the Rust `Renderer` with its
// `engine` field swapped for the FFI binding shown in INTEGRATION.md.
#pragma once
#include <string>
#include <vector>

#include "../../../include/vello_hip.h"
#include "scene.hpp"

namespace vello {

enum class AaConfig : uint32_t { Area = 0, Msaa8 = 1, Msaa16 = 2 };

struct RenderParams {
    Color base_color{0.f, 0.f, 0.f, 1.f};
    uint32_t width = 0, height = 0;
    AaConfig antialiasing_method = AaConfig::Area;
};

struct RendererOptions {
    int device = 0;
    uint32_t antialiasing_support = VELLO_HIP_AA_MASK_ALL;  // AaSupport::all()
    vello_hip_capacities capacities{};                      // zero = reference pool sizes
};

class Renderer {
  public:
    // Renderer::new: returns nullptr and fills *err when no gfx950 device is usable.
    static Renderer *create(const RendererOptions &options, std::string *err);
    ~Renderer();
    // texture: linear RGBA8 (Rgba8Unorm) buffer of `stride` bytes per row, device memory when
    // is_device.  Returns a VELLO_HIP_* code; error() describes the last failure.
    int render_to_texture(const Scene &scene, void *texture, size_t stride, bool is_device, const RenderParams &params);
    const std::string &error() const { return error_; }
    vello_hip_ctx *engine() { return ctx_; }
    const vello_hip_bump &last_bump() const { return bump_; }

  private:
    Renderer() = default;
    vello_hip_ctx *ctx_ = nullptr;
    vello_encoding::Resolver resolver_;
    std::vector<uint8_t> packed_;
    vello_hip_bump bump_{};
    std::string error_;
};

}  // namespace vello
