// RampCache, ImageCache and Resolver::resolve: the late-bound resource side of vello_encoding
// (ramp_cache.rs, image_cache.rs, resolve.rs:172-393).  Declarations in encoding.hpp.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "encoding.hpp"

namespace vello_encoding {

// ---------------- RampCache (ramp_cache.rs:46-117) ----------------
namespace {

struct Rgba {
    float c[4];
};

uint32_t fbits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

std::vector<uint32_t> ramp_key(InterpolationAlphaSpace space, const ColorStop *stops, size_t n) {
    std::vector<uint32_t> k;
    k.reserve(1 + n * 5);
    k.push_back((uint32_t)space);
    for (size_t i = 0; i < n; i++) {
        k.push_back(fbits(stops[i].offset));
        k.push_back(fbits(stops[i].color.r));
        k.push_back(fbits(stops[i].color.g));
        k.push_back(fbits(stops[i].color.b));
        k.push_back(fbits(stops[i].color.a));
    }
    return k;
}

// make_ramp (ramp_cache.rs:119-155)
void make_ramp(const ColorStop *stops, size_t n, InterpolationAlphaSpace space, uint32_t *out) {
    auto col = [&](size_t j) { return Rgba{{stops[j].color.r, stops[j].color.g, stops[j].color.b, stops[j].color.a}}; };
    float last_u = 0.0f;
    Rgba last_c = col(0);
    float this_u = last_u;
    Rgba this_c = last_c;
    size_t j = 0;
    for (size_t i = 0; i < RampCache::N_SAMPLES; i++) {
        float u = (float)i / (float)(RampCache::N_SAMPLES - 1);
        while (u > this_u) {
            last_u = this_u;
            last_c = this_c;
            if (j + 1 < n) {
                this_u = stops[j + 1].offset;
                this_c = col(j + 1);
                j += 1;
            } else {
                break;
            }
        }
        float du = this_u - last_u;
        Rgba c;
        if (du < 1e-9f) {
            c = this_c;
        } else {
            float t = (u - last_u) / du;
            if (space == InterpolationAlphaSpace::Premultiplied) {
                // AlphaColor::lerp (color crate): interpolate premultiplied, then un-premultiply
                Rgba a = last_c, b = this_c, r;
                for (int k = 0; k < 3; k++) {
                    a.c[k] *= a.c[3];
                    b.c[k] *= b.c[3];
                }
                for (int k = 0; k < 4; k++) r.c[k] = a.c[k] + t * (b.c[k] - a.c[k]);
                if (r.c[3] == 0.0f) {
                    c = Rgba{{0.f, 0.f, 0.f, 0.f}};
                } else {
                    float inv = 1.0f / r.c[3];
                    c = Rgba{{r.c[0] * inv, r.c[1] * inv, r.c[2] * inv, r.c[3]}};
                }
            } else {
                for (int k = 0; k < 4; k++) c.c[k] = last_c.c[k] + (this_c.c[k] - last_c.c[k]) * t;
            }
        }
        out[i] = Color{c.c[0], c.c[1], c.c[2], c.c[3]}.premul_rgba8();
    }
}

}  // namespace

void RampCache::maintain() {
    epoch_ += 1;
    if (map_.size() > RETAINED_COUNT) {
        map_.erase(std::remove_if(map_.begin(), map_.end(), [](const Entry &e) { return e.id >= RETAINED_COUNT; }), map_.end());
        data_.resize(RETAINED_COUNT * N_SAMPLES);
    }
}

uint32_t RampCache::add(InterpolationAlphaSpace space, const ColorStop *stops, size_t n) {
    std::vector<uint32_t> key = ramp_key(space, stops, n);
    for (Entry &e : map_) {
        if (e.key == key) {
            e.epoch = epoch_;
            return e.id;
        }
    }
    auto append = [&]() {
        uint32_t id = (uint32_t)(data_.size() / N_SAMPLES);
        data_.resize(data_.size() + N_SAMPLES);
        make_ramp(stops, n, space, data_.data() + (size_t)id * N_SAMPLES);
        map_.push_back(Entry{std::move(key), id, epoch_});
        return id;
    };
    if (map_.size() < RETAINED_COUNT) return append();
    for (size_t i = 0; i < map_.size(); i++) {
        if (map_[i].epoch + 2 < epoch_) {
            uint32_t id = map_[i].id;
            map_.erase(map_.begin() + (long)i);
            make_ramp(stops, n, space, data_.data() + (size_t)id * N_SAMPLES);
            map_.push_back(Entry{std::move(key), id, epoch_});
            return id;
        }
    }
    return append();
}

// ---------------- ImageCache (image_cache.rs:61-210) ----------------
bool ImageCache::Shelves::alloc(uint32_t w, uint32_t h, uint32_t *x, uint32_t *y) {
    if (w > size || h > size) return false;
    if (shelf_x + w > size) {  // next shelf
        shelf_y += shelf_h;
        shelf_h = 0;
        shelf_x = 0;
    }
    if (shelf_y + h > size) return false;
    *x = shelf_x;
    *y = shelf_y;
    shelf_x += w;
    shelf_h = std::max(shelf_h, h);
    return true;
}

void ImageCache::begin_resolve() {
    generation_ += 1;
    evicted_in_resolve_ = 0;
    uploads_.clear();
}

void ImageCache::restart_resolve_pass() {
    uploads_.clear();
    for (Resident &r : resident_)
        if (r.last_used_generation == generation_) r.last_used_generation = generation_ - 1;
}

bool ImageCache::is_resident(const ImageData &image) const {
    for (const Resident &r : resident_)
        if (r.image.id == image.id) return true;
    return false;
}

bool ImageCache::get_or_insert(const ImageData &image, uint32_t *x, uint32_t *y) {
    if (!shelves_init_) {
        shelves_ = Shelves{size_};
        shelves_init_ = true;
    }
    for (Resident &r : resident_) {
        if (r.image.id == image.id) {
            *x = r.x;
            *y = r.y;
            if (r.last_used_generation != generation_) {
                r.last_used_generation = generation_;
                if (r.dirty) uploads_.push_back(ImageUpload{r.image, r.x, r.y});
            }
            return true;
        }
    }
    if (!shelves_.alloc(image.width, image.height, x, y)) return false;
    resident_.push_back(Resident{image, *x, *y, true, generation_});
    uploads_.push_back(ImageUpload{image, *x, *y});
    return true;
}

void ImageCache::finish_resolve() {
    for (Resident &r : resident_)
        if (r.last_used_generation == generation_) r.dirty = false;
}

void ImageCache::mark_dirty(const ImageData &image) {
    for (Resident &r : resident_)
        if (r.image.id == image.id) r.dirty = true;
}

// Places `keep` (in blob-id order, as image_cache.rs:184-188 does) into a fresh atlas of side `size`; every entry
// becomes dirty because it moved.  All or nothing.
bool ImageCache::repack(uint32_t size, const std::vector<Resident> &keep) {
    std::vector<Resident> sorted = keep;
    std::sort(sorted.begin(), sorted.end(), [](const Resident &a, const Resident &b) { return a.image.id < b.image.id; });
    Shelves shelves{size};
    for (Resident &r : sorted) {
        if (!shelves.alloc(r.image.width, r.image.height, &r.x, &r.y)) return false;
        r.dirty = true;
    }
    resident_ = std::move(sorted);
    shelves_ = shelves;
    shelves_init_ = true;
    size_ = size;
    resized_ = true;
    return true;
}

bool ImageCache::repack_to_size(uint32_t size) { return repack(size, resident_); }

// image_cache.rs:101-111: double the side until everything that is resident fits
bool ImageCache::bump_size() {
    for (uint64_t new_size = (uint64_t)size_ * 2u; new_size <= max_size_; new_size *= 2u) {
        if (repack_to_size((uint32_t)new_size)) {
            uploads_.clear();
            return true;
        }
    }
    return false;
}

// image_cache.rs:167-182: residents no resolve has used for EVICT_AFTER_GENERATIONS generations make room
bool ImageCache::evict_stale_entries() {
    if (evicted_in_resolve_ != 0u) return false;
    if (generation_ < EVICT_AFTER_GENERATIONS) return false;
    const uint64_t stale_before = generation_ - EVICT_AFTER_GENERATIONS;
    std::vector<Resident> keep;
    for (const Resident &r : resident_)
        if (r.last_used_generation >= stale_before) keep.push_back(r);
    const uint32_t n_evicted = (uint32_t)(resident_.size() - keep.size());
    if (n_evicted == 0u) return false;
    if (!repack(size_, keep)) return false;  // cannot happen for a subset in practice; the residency is untouched if it does
    uploads_.clear();
    evicted_in_resolve_ = n_evicted;
    return true;
}

// ---------------- Resolver::resolve (resolve.rs:172-393, glyph runs not restated) ----------------
Resolved Resolver::resolve(const Encoding &encoding, std::vector<uint8_t> &data) {
    Resolved out;
    const Resources &resources = encoding.resources;
    if (resources.patches.empty()) {
        out.layout = resolve_solid_paths_only(encoding, data);
        return out;
    }
    // resolve_patches (resolve.rs:395-505)
    struct ResolvedPatch {
        Patch::Kind kind;
        size_t draw_data_offset;
        uint32_t ramp_id;
        Extend extend;
        size_t image_index;
    };
    struct PendingImage {
        ImageData image;
        bool placed;
        uint32_t x, y;
    };
    ramp_cache_.maintain();
    image_cache_.begin_resolve();
    std::vector<ResolvedPatch> patches;
    std::vector<PendingImage> pending;
    bool has_images = false;
    for (const Patch &p : resources.patches) {
        if (p.kind == Patch::Kind::Ramp) {
            uint32_t ramp_id = ramp_cache_.add(p.interpolation_alpha_space, resources.color_stops.data() + p.stops_begin,
                                               p.stops_end - p.stops_begin);
            patches.push_back({Patch::Kind::Ramp, p.draw_data_offset, ramp_id, p.extend, 0});
        } else {
            has_images = true;
            patches.push_back({Patch::Kind::Image, p.draw_data_offset, 0u, Extend::Pad, pending.size()});
            pending.push_back({p.image, false, 0u, 0u});
        }
    }
    // resolve_pending_images (resolve.rs:507-541): place every image; under pressure first drop what no recent frame
    // used, then grow the atlas, until everything fits or the maximum size is hit (such an image is not drawn).
    // Both remedies move the residents here, so both restart the placement pass.
    for (bool restart = true; restart;) {
        restart = false;
        image_cache_.restart_resolve_pass();
        for (PendingImage &pi : pending) {
            pi.placed = image_cache_.get_or_insert(pi.image, &pi.x, &pi.y);
            if (pi.placed) continue;
            if ((image_cache_.can_fit_image(pi.image) && image_cache_.evict_stale_entries()) || image_cache_.bump_size()) {
                restart = true;
                break;
            }
        }
    }
    data.clear();
    Layout layout{};
    layout.n_paths = encoding.n_paths;
    layout.n_clips = encoding.n_clips;
    auto words = [&]() { return (uint32_t)(data.size() / 4); };
    auto push_u32s = [&](const uint32_t *p, size_t n) {
        const uint8_t *b = reinterpret_cast<const uint8_t *>(p);
        data.insert(data.end(), b, b + n * 4);
    };
    size_t n_path_tags = encoding.path_tags.size() + encoding.n_open_clips;
    size_t path_tag_padded = (n_path_tags + 1023u) / 1024u * 1024u;
    layout.path_tag_base = words();
    data.insert(data.end(), encoding.path_tags.begin(), encoding.path_tags.end());
    for (uint32_t i = 0; i < encoding.n_open_clips; i++) data.push_back(PathTag::PATH);
    data.resize(path_tag_padded, 0);
    layout.path_data_base = words();
    push_u32s(encoding.path_data.data(), encoding.path_data.size());
    layout.draw_tag_base = words();
    uint32_t bds = 0;
    for (uint32_t t : encoding.draw_tags) bds += DrawTag::info_size(t);
    layout.bin_data_start = bds;
    push_u32s(encoding.draw_tags.data(), encoding.draw_tags.size());
    for (uint32_t i = 0; i < encoding.n_open_clips; i++) {
        uint32_t t = DrawTag::END_CLIP;
        push_u32s(&t, 1);
    }
    // draw data with the patches applied (resolve.rs:277-321)
    layout.draw_data_base = words();
    {
        size_t pos = 0;
        const std::vector<uint32_t> &stream = encoding.draw_data;
        for (const ResolvedPatch &p : patches) {
            if (pos < p.draw_data_offset) push_u32s(stream.data() + pos, p.draw_data_offset - pos);
            if (p.kind == Patch::Kind::Ramp) {
                uint32_t index_mode = (p.ramp_id << 2) | (uint32_t)p.extend;
                push_u32s(&index_mode, 1);
                pos = p.draw_data_offset + 1;
            } else {
                const PendingImage &pi = pending[p.image_index];
                if (pi.placed) {
                    uint32_t xy = (pi.x << 16) | pi.y;
                    push_u32s(&xy, 1);
                    pos = p.draw_data_offset + 1;
                } else {
                    // no room in the atlas: zero the dimensions so nothing is sampled (resolve.rs:309-316)
                    uint32_t z[2] = {0u, 0u};
                    push_u32s(z, 2);
                    pos = p.draw_data_offset + 2;
                }
            }
        }
        if (pos < stream.size()) push_u32s(stream.data() + pos, stream.size() - pos);
    }
    layout.transform_base = words();
    push_u32s(reinterpret_cast<const uint32_t *>(encoding.transforms.data()), encoding.transforms.size() * 6);
    layout.style_base = words();
    push_u32s(reinterpret_cast<const uint32_t *>(encoding.styles.data()), encoding.styles.size() * 2);
    layout.n_draw_objects = layout.n_paths;
    out.layout = layout;
    out.ramps = ramp_cache_.data().data();
    out.n_ramps = ramp_cache_.height();
    if (has_images) {
        out.atlas_size = image_cache_.size();
        out.atlas_resized = image_cache_.resized();
        image_cache_.clear_resized();
        out.uploads = &image_cache_.uploads();
        out.evicted = image_cache_.evicted();
    }
    image_cache_.finish_resolve();
    return out;
}

}  // namespace vello_encoding
