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
#include <memory>
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

// peniko::{Extend, ColorStop, InterpolationAlphaSpace, ImageFormat, ImageAlphaType, ImageQuality}: the enum values
// are the ones the shaders decode (fine.wgsl:829-875, :26-28)
enum class Extend : uint32_t { Pad = 0, Repeat = 1, Reflect = 2 };
enum class InterpolationAlphaSpace : uint32_t { Premultiplied = 0, Unpremultiplied = 1 };
enum class ImageFormat : uint32_t { Rgba8 = 0, Bgra8 = 1 };
enum class ImageAlphaType : uint32_t { Alpha = 0, AlphaPremultiplied = 1 };
enum class ImageQuality : uint32_t { Low = 0, Medium = 1, High = 2 };
struct ColorStop {
    float offset;
    Color color;
    ColorStop multiply_alpha(float alpha) const { return {offset, color.multiply_alpha(alpha)}; }
};

// peniko::ImageData: `id` plays the role of Blob::id() (the atlas cache key, image_cache.rs:101)
struct ImageData {
    uint64_t id = 0;
    uint32_t width = 0, height = 0;
    ImageFormat format = ImageFormat::Rgba8;
    ImageAlphaType alpha_type = ImageAlphaType::Alpha;
    std::shared_ptr<const std::vector<uint8_t>> data;  // width * height * 4 bytes
};
struct ImageSampler {
    Extend x_extend = Extend::Pad, y_extend = Extend::Pad;
    ImageQuality quality = ImageQuality::Medium;
    float alpha = 1.0f;
};
struct ImageBrush {
    ImageData image;
    ImageSampler sampler;
};

// draw.rs:117-186
struct DrawLinearGradient {
    uint32_t index;
    float p0[2], p1[2];
};
struct DrawRadialGradient {
    uint32_t index;
    float p0[2], p1[2], r0, r1;
};
struct DrawSweepGradient {
    uint32_t index;
    float p0[2], t0, t1;
};

// peniko::Gradient
struct Gradient {
    enum class Kind { Linear, Radial, Sweep } kind = Kind::Linear;
    double p0[2] = {0, 0}, p1[2] = {0, 0};  // linear: start/end; radial: start_center/end_center; sweep: center = p0
    float r0 = 0, r1 = 0;                   // radial radii
    float start_angle = 0, end_angle = 0;   // sweep
    Extend extend = Extend::Pad;
    InterpolationAlphaSpace interpolation_alpha_space = InterpolationAlphaSpace::Premultiplied;
    std::vector<ColorStop> stops;
};

// resolve.rs:563-590 (Patch), encoding.rs:560-590 (Resources); glyph runs are not restated
struct Patch {
    enum class Kind { Ramp, Image } kind;
    size_t draw_data_offset;  // in u32 words of draw_data
    size_t stops_begin = 0, stops_end = 0;
    Extend extend = Extend::Pad;
    InterpolationAlphaSpace interpolation_alpha_space = InterpolationAlphaSpace::Premultiplied;
    ImageData image;
};
struct Resources {
    std::vector<Patch> patches;
    std::vector<ColorStop> color_stops;
    void reset() {
        patches.clear();
        color_stops.clear();
    }
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
    Resources resources;
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
    // encoding.rs:352-484
    void encode_gradient(const Gradient &gradient, float alpha);
    void encode_linear_gradient(DrawLinearGradient gradient, const std::vector<ColorStop> &stops, float alpha, Extend extend,
                                InterpolationAlphaSpace space);
    void encode_radial_gradient(DrawRadialGradient gradient, const std::vector<ColorStop> &stops, float alpha, Extend extend,
                                InterpolationAlphaSpace space);
    void encode_sweep_gradient(DrawSweepGradient gradient, const std::vector<ColorStop> &stops, float alpha, Extend extend,
                               InterpolationAlphaSpace space);
    void encode_image(const ImageBrush &brush, float alpha);
    void encode_blurred_rounded_rect(uint32_t premul_rgba8, float width, float height, float radius, float std_dev);
    void encode_begin_clip(const DrawBeginClip &p);
    void encode_end_clip();
    void swap_last_path_tags();

  private:
    void encode_style(const Style &s);
    enum class RampStops { Empty, One, Many };
    RampStops add_ramp(const std::vector<ColorStop> &stops, float alpha, Extend extend, InterpolationAlphaSpace space, Color *one);
};

struct Layout {
    uint32_t n_draw_objects, n_paths, n_clips, bin_data_start;
    uint32_t path_tag_base, path_data_base, draw_tag_base, draw_data_base, transform_base, style_base;
    uint32_t path_tags_size() const { return (path_data_base - path_tag_base) * 4u; }
};
static_assert(sizeof(Layout) == 40, "Layout");

// resolve.rs:107-154
Layout resolve_solid_paths_only(const Encoding &encoding, std::vector<uint8_t> &packed);

// ramp_cache.rs:12-155.  make_ramp interpolates with the `color` crate (0.3.x, not in the reference tree):
// AlphaColor<Srgb>::lerp = premultiply, component-wise a + t*(b-a), un-premultiply; restated here.
class RampCache {
  public:
    static constexpr size_t N_SAMPLES = 512, RETAINED_COUNT = 64;
    void maintain();
    uint32_t add(InterpolationAlphaSpace space, const ColorStop *stops, size_t n);
    const std::vector<uint32_t> &data() const { return data_; }
    uint32_t height() const { return (uint32_t)(data_.size() / N_SAMPLES); }

  private:
    struct Entry {
        std::vector<uint32_t> key;  // bit pattern of (space, offset, r, g, b, a)*
        uint32_t id;
        uint64_t epoch;
    };
    uint64_t epoch_ = 0;
    std::vector<Entry> map_;
    std::vector<uint32_t> data_;
};

// image_cache.rs: residency map keyed by image id + an atlas allocator.  The reference packs with guillotiere
// 0.7 (not in the tree); the positions are free for the host to choose (the kernels only read the xy the
// Resolver patches into DrawImage), so this mirror uses a shelf packer with the same sizes and growth rule
// (1024 -> 8192, doubling, image_cache.rs:9-11, :76-86).
struct ImageUpload {
    ImageData image;
    uint32_t x, y;
};
// image_cache.rs:37-210.  Residency, dirtiness and staleness follow the reference; the allocator does not: the
// reference delegates to the guillotiere crate (not vendored), here a shelf packer places images, and because a shelf
// packer cannot free a single rectangle an eviction repacks the survivors (they are re-uploaded, like after a growth).
// Atlas positions are late-bound into the draw data at every resolve, so they are not part of the parity surface.
class ImageCache {
  public:
    static constexpr uint32_t DEFAULT_ATLAS_SIZE = 1024, MAX_ATLAS_SIZE = 8192;
    static constexpr uint64_t EVICT_AFTER_GENERATIONS = 2;
    ImageCache() = default;
    ImageCache(uint32_t initial_size, uint32_t max_size) : size_(initial_size), max_size_(max_size) {}
    void begin_resolve();                  // a new generation; nothing queued for upload, nothing evicted yet
    void restart_resolve_pass();           // placement starts over (after a growth / eviction moved the residents)
    bool get_or_insert(const ImageData &image, uint32_t *x, uint32_t *y);
    void finish_resolve();                 // what this generation used is clean now
    void mark_dirty(const ImageData &image);
    bool can_fit_image(const ImageData &image) const { return image.width <= size_ && image.height <= size_; }
    bool evict_stale_entries();            // at most once per resolve; true if anything was evicted
    bool bump_size();
    bool repack_to_size(uint32_t size);    // all or nothing: a failed repack leaves the residency unchanged
    uint32_t size() const { return size_; }
    uint32_t evicted() const { return evicted_in_resolve_; }
    size_t n_resident() const { return resident_.size(); }
    bool is_resident(const ImageData &image) const;
    const std::vector<ImageUpload> &uploads() const { return uploads_; }
    bool resized() const { return resized_; }
    void clear_resized() { resized_ = false; }

  private:
    struct Resident {
        ImageData image;
        uint32_t x, y;
        bool dirty;
        uint64_t last_used_generation;
    };
    struct Shelves {
        uint32_t size, shelf_y = 0, shelf_h = 0, shelf_x = 0;
        bool alloc(uint32_t w, uint32_t h, uint32_t *x, uint32_t *y);
    };
    bool repack(uint32_t size, const std::vector<Resident> &keep);
    uint32_t size_ = DEFAULT_ATLAS_SIZE, max_size_ = MAX_ATLAS_SIZE;
    Shelves shelves_{DEFAULT_ATLAS_SIZE};
    bool shelves_init_ = false;
    uint64_t generation_ = 0;
    uint32_t evicted_in_resolve_ = 0;
    std::vector<Resident> resident_;
    std::vector<ImageUpload> uploads_;
    bool resized_ = true;  // the atlas texture has to be (re)created before the first upload
};

// Resolver::resolve (resolve.rs:172-393) for encodings with ramp / image patches.
struct Resolved {
    Layout layout;
    const uint32_t *ramps = nullptr;  // 512 texels per ramp
    uint32_t n_ramps = 0;
    uint32_t atlas_size = 0;          // square atlas side; 0 when the scene has no images
    bool atlas_resized = false;       // the atlas must be re-created: grown, or repacked after an eviction (every image
                                      // this frame samples is in `uploads`)
    uint32_t evicted = 0;             // stale residents dropped by this resolve (image_cache.rs:19-22)
    const std::vector<ImageUpload> *uploads = nullptr;
};
class Resolver {
  public:
    Resolver() = default;
    Resolver(uint32_t atlas_initial_size, uint32_t atlas_max_size) : image_cache_(atlas_initial_size, atlas_max_size) {}
    Resolved resolve(const Encoding &encoding, std::vector<uint8_t> &packed);
    // resolve.rs:173-179: the next resolve that uses `image` uploads it again (its pixels changed in place)
    void mark_image_dirty(const ImageData &image) { image_cache_.mark_dirty(image); }
    const ImageCache &image_cache() const { return image_cache_; }

  private:
    RampCache ramp_cache_;
    ImageCache image_cache_;
};

}  // namespace vello_encoding
