// The README example of the reference (a filled shape rendered through `Renderer::render_to_texture`), in C++ on this
// repository's host mirror: build a Scene, render it on the MI355X, write a binary PPM.
//
//   g++ -std=c++17 -O2 examples/hello_gradient.cpp -I . -L vello_amd/lib -lvello_hip -Wl,-rpath,$PWD/vello_amd/lib -o hello
//   ./hello out.ppm
//
// Reference counterpart: README.md "Getting started" / examples/simple (vello/src/lib.rs:432-515).
#include <cstdio>
#include <string>
#include <vector>

#include "vello_amd/csrc/host/renderer.hpp"

int main(int argc, char **argv) {
    using namespace vello;
    const uint32_t width = 512, height = 512;

    Scene scene;
    // a sweep-gradient disc, a translucent stroked ring clipped to its own outline, and a blurred shadow
    vello_encoding::Gradient sweep;
    sweep.kind = vello_encoding::Gradient::Kind::Sweep;
    sweep.p0[0] = 256.0;
    sweep.p0[1] = 256.0;
    sweep.start_angle = 0.0f;
    sweep.end_angle = 6.2831853f;
    sweep.stops = {{0.0f, Color{1.f, 0.f, 0.f, 1.f}}, {0.33f, Color{0.f, 1.f, 0.f, 1.f}}, {0.66f, Color{0.f, 0.f, 1.f, 1.f}},
                   {1.0f, Color{1.f, 0.f, 0.f, 1.f}}};
    scene.draw_blurred_rounded_rect(Affine::identity(), kurbo::Rect{96.0, 112.0, 416.0, 432.0}, Color{0.f, 0.f, 0.f, 0.6f}, 160.0, 12.0);
    scene.fill(Fill::NonZero, Affine::identity(), Brush(sweep), std::nullopt,
               kurbo::path_elements(kurbo::Circle{{256.0, 256.0}, 150.0}, 0.1));
    kurbo::Stroke ring;
    ring.width = 24.0;
    scene.push_clip_layer(ring, Affine::identity(), kurbo::path_elements(kurbo::Circle{{256.0, 256.0}, 190.0}, 0.1));
    scene.fill(Fill::NonZero, Affine::identity(), Color{1.f, 1.f, 1.f, 0.7f}, kurbo::path_elements(kurbo::Rect{0.0, 0.0, 512.0, 512.0}, 0.1));
    scene.pop_layer();

    std::string err;
    Renderer *renderer = Renderer::create(RendererOptions{}, &err);
    if (!renderer) {
        std::fprintf(stderr, "no renderer: %s\n", err.c_str());  // no MI355X: there is no CPU fallback
        return 2;
    }
    std::vector<uint8_t> rgba((size_t)width * height * 4u);
    RenderParams params;
    params.base_color = Color{0.1f, 0.1f, 0.12f, 1.f};
    params.width = width;
    params.height = height;
    params.antialiasing_method = AaConfig::Msaa16;
    int r = renderer->render_to_texture(scene, rgba.data(), (size_t)width * 4u, /*is_device=*/false, params);
    if (r != VELLO_HIP_OK) {
        std::fprintf(stderr, "render failed (%d): %s\n", r, renderer->error().c_str());
        delete renderer;
        return 1;
    }
    if (argc > 1) {
        FILE *f = std::fopen(argv[1], "wb");
        if (!f) return 1;
        std::fprintf(f, "P6\n%u %u\n255\n", width, height);
        for (size_t i = 0; i < (size_t)width * height; i++) std::fwrite(&rgba[i * 4], 1, 3, f);
        std::fclose(f);
    }
    const vello_hip_bump &b = renderer->last_bump();
    std::printf("rendered %ux%u: %u lines, %u tile crossings, %u segments\n", width, height, b.lines, b.seg_counts, b.segments);
    delete renderer;
    return 0;
}
