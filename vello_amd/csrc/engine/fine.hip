// fine: per-tile rasterization + compositing -> RGBA8.
// Reference: vello_shaders/shader/fine.wgsl (area AA :1005-1059, MSAA :146-709, command
// interpreter :1064-1398), shared/blend.wgsl:147-319 (vello/src/render.rs:560-629).
//
// gfx950 design: the reference workgroup is 4x16 = 64 invocations, i.e. exactly one wave64, so one
// wave owns one 16x16 tile and every cross-lane step of the reference becomes a wave operation:
//  * the per-batch Hillis-Steele scan of pixel counts is a 6-step shuffle scan;
//  * each batch of 64 segments is staged into LDS once with coalesced 24-B loads and all later
//    passes (area loop / pixel walk) read it from LDS (the reference re-reads global memory per
//    lane: fine.wgsl:1018, :226);
//  * the winding accumulators (sh_samples etc.) are LDS atomics on the packed 8-bit counters of
//    the reference, kept bit-identical so the MSAA coverage is integer-exact;
//  * the half-plane mask LUT is a persistent device buffer (the reference re-uploads it every
//    frame: render.rs:583-591), read through L1/L2;
//  * each lane owns 4 horizontally adjacent pixels and stores them as one 16-byte write.
#include "engine.h"

namespace vk {

namespace {

constexpr uint32_t PIXELS_PER_THREAD = 4;
constexpr int GRADIENT_WIDTH = 512;
constexpr uint32_t LUMINANCE_MASK_LAYER = 0x10000u;

// Measurement build only (make -C vello_amd/csrc EXTRA=-DVELLO_FINE_PROF, scripts/fine_prof.py): clock64 per phase of a
// tile's wave, written to the tail of the blend-spill pool when the tile is done.  Without the macro FineProf is empty
// and every mark() / count() is nothing: the product kernel's code is the same with or without these lines.
enum {
    FP_INTERP = 0,      // command decode, window loads, everything not listed below
    FP_BATCH_SCAN,      // ms_build_batch: scan of the window, slots, segment addresses
    FP_BATCH_SEGS,      // ... segment loads (the global round trip), setup, counts
    FP_BATCH_ITEMS,     // ... one record per crossing (mask LUT loads)
    FP_FILL_APPLY,      // ms_fill_from_batch: records -> counter atomics (incl. the slot's parameters from LDS)
    FP_FILL_PREFIX,     // ... winding prefix sums, expected_zero, exchange through LDS
    FP_FILL_SPARSE,     // ... one lane per record evaluates its pixel
    FP_FILL_RESTORE,    // ... counters back to the cleared value, coverage picked up
    FP_FILL_EVENODD,    // even-odd fills of a batch (dense path)
    FP_FILL_UNBATCHED,  // fill_path_ms
    FP_BLEND,           // CMD_COLOR src-over
    FP_RARE,            // rare_command: BEGIN_CLIP (and anything not listed below)
    FP_RARE_END_CLIP,   // ... END_CLIP: the blend (blend_mix_compose, luminance masks)
    FP_RARE_GRAD,       // ... LIN / RAD / SWEEP gradients
    FP_RARE_IMAGE,      // ... IMAGE, BLUR_RECT
    FP_N_FILLS, FP_N_BATCHES, FP_N_ITEMS, FP_N_WORDS,  // counts: fills, batches, crossing records, command words
    FP_N_RARE,          // count: rare commands
    FP_N_SIMPLE,        // count: fills taken by ms_fill_simple
    FP_SLOTS
};
#ifdef VELLO_FINE_PROF
struct FineProf {
    uint32_t acc[FP_SLOTS];
    long long prev;
    __device__ __forceinline__ void start() {
        for (int i = 0; i < FP_SLOTS; i++) acc[i] = 0u;
        prev = clock64();
    }
    __device__ __forceinline__ void mark(int k) {
        const long long t = clock64();
        acc[k] += (uint32_t)(t - prev);
        prev = t;
    }
    __device__ __forceinline__ void count(int k, uint32_t n) { acc[k] += n; }
    __device__ __forceinline__ void store(uint32_t *blend_spill, uint32_t blend_size, uint32_t n_tiles, uint32_t tile_ix, uint32_t lane) {
        if (blend_size < n_tiles * FP_SLOTS) return;
        uint32_t *dst = blend_spill + (blend_size - n_tiles * FP_SLOTS) + tile_ix * FP_SLOTS;
        if (lane == 0u)
            for (int i = 0; i < FP_SLOTS; i++) dst[i] = acc[i];
    }
};
#else
struct FineProf {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void count(int, uint32_t) {}
    __device__ __forceinline__ void store(uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t) {}
};
#endif

// Measurement build only (EXTRA=-DVELLO_FINE_TIMELINE, scripts/fine_timeline.py): every wave of k_fine logs when it started
// and ended (wall clock, 100 MHz), where it ran (HW_ID) and what it was (tile / slice / slice + compositing pass) into the
// tail of the blend-spill pool: the occupancy of the chip over the launch.  Nothing without the macro.
#ifdef VELLO_FINE_TIMELINE
constexpr uint32_t TL_MAX = 40000u;
struct FineTimeline {
    uint32_t t0;
    __device__ __forceinline__ void start() { t0 = (uint32_t)wall_clock64(); }
    __device__ __forceinline__ void log(uint32_t *blend_spill, uint32_t blend_size, uint32_t lane, uint32_t kind, uint32_t fills, uint32_t tile_ix) {
        if (blend_size < 6u * TL_MAX + 1u || lane != 0u) return;
        const uint32_t slot = atomicAdd(&blend_spill[blend_size - 1u], 1u);
        if (slot >= TL_MAX) return;
        uint32_t *dst = blend_spill + (blend_size - 1u - 6u * TL_MAX) + slot * 6u;
        dst[0] = t0;
        dst[1] = (uint32_t)wall_clock64();
        dst[2] = __builtin_amdgcn_s_getreg((23 << 0) | (0 << 6) | (31 << 11));  // HW_ID, all 32 bits
        dst[3] = kind | (fills << 8);
        dst[4] = tile_ix;
        dst[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    }
};
#else
struct FineTimeline {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void log(uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) {}
};
#endif

struct vec4 {
    float x, y, z, w;
};
__device__ __forceinline__ vec4 operator*(vec4 a, float s) { return vec4{a.x * s, a.y * s, a.z * s, a.w * s}; }
__device__ __forceinline__ vec4 operator+(vec4 a, vec4 b) { return vec4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

// (float)b / 255.0f for b in [0, 255] in three operations: q = b*r, one fma residual, one fma correction.  Equal to
// the IEEE division for all 256 inputs (checked exhaustively in tests/test_emu_parity.py); the compiler's generic
// division expansion costs 11 VALU per channel and CMD_COLOR unpacks four channels per command.
__device__ __forceinline__ float unorm8_to_f32(uint32_t b) {
    const float r = 1.0f / 255.0f;
    const float x = (float)b;
    const float q = x * r;
    const float e = fmaf(-q, 255.0f, x);
    return fmaf(e, r, q);
}
__device__ __forceinline__ vec4 unpack4x8unorm(uint32_t u) {
    return vec4{unorm8_to_f32(u & 0xffu), unorm8_to_f32((u >> 8) & 0xffu), unorm8_to_f32((u >> 16) & 0xffu), unorm8_to_f32(u >> 24)};
}
__device__ __forceinline__ uint32_t unorm8(float e) { return (uint32_t)floorf(0.5f + 255.0f * clampf(e, 0.0f, 1.0f)); }
__device__ __forceinline__ uint32_t pack4x8unorm(vec4 c) {
    return unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
}

struct CmdFill {
    uint32_t size_and_rule, seg_data;
    int32_t backdrop;
};

// Per-segment line setup of a batch (ms_setup), live only while ms_build_batch runs: shares its storage with the
// staged Segment records of the one-fill-at-a-time paths (never both at once).
struct SegSetupLds {
    float a[64], b[64], mask_row[64];
    int32_t x0i[64], y0i[64];
    uint32_t flags[64], edge_masks[64];
    uint32_t unused[64];  // (PixelLds, which shares this storage later, is this long)
};
// Per-pixel exchange of ms_fill_from_batch (live while the fills of a staged batch are resolved; written after
// ms_build_batch is done with SegSetupLds, whose storage it shares): `pw` = a lane's four packed x-winding prefixes,
// `area` = the coverage, indexed by pixel, and `zero_at` = per staged fill and pixel row the byte an x-winding prefix
// holds where the winding number is zero, replicated into the four bytes of a word (ms_build_batch, end).
struct alignas(16) PixelLds {
    uint32_t pw[64];
    float area[256];
    uint32_t zero_at[12][16];
};
#ifdef VELLO_SIMT_EMU
#define FINE_WAVE_ANY(c) (__ballot(c) != 0ull)
#else
#define FINE_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#endif
#ifndef VK_FINE_REGS
#define VK_FINE_REGS 0  // 1: a simple fill's touched pixels from their records in registers where no pixel holds three (prototype)
#endif
struct FineShared {
    // INVARIANT: `seg` is written only by fill_path_area / fill_path_ms, and those run only while NO batch is staged
    // (k_fine's loop: batch_pos == batch_n, and ms_build_batch returned 0 for the fill).  px.zero_at is written once per batch and
    // must survive every fill of the batch; `seg` covers px.pw, px.area and rows 0-3 of px.zero_at (asserted below), so a use of
    // `seg` between ms_build_batch and the batch's last ms_fill_from_batch would corrupt the coverage of slots 0-3.  (zero_at
    // cannot move out of the union: 768 B more per wave is the difference between 16 and 15 waves of LDS per CU.)
    union {
        Segment seg[64];
        SegSetupLds su;
        PixelLds px;
    };
    uint32_t count[64];
    uint32_t winding_y[4];
    uint32_t winding_y_prefix[4];
    uint32_t winding[64];
};

// ---------------- area AA (fine.wgsl:1005-1059) ----------------
__device__ void fill_path_area(FineShared &sh, const Segment *__restrict__ segments, CmdFill fill, uint32_t lane, float (&area)[4],
                               const Segment &first) {
    const uint32_t n_segs = fill.size_and_rule >> 1;
    const bool even_odd = (fill.size_and_rule & 1u) != 0u;
    const float xy_x = (float)((lane & 3u) * PIXELS_PER_THREAD);
    const float xy_y = (float)(lane >> 2);
    const float backdrop_f = (float)fill.backdrop;
#pragma unroll
    for (int k = 0; k < 4; k++) area[k] = backdrop_f;
    for (uint32_t base = 0; base < n_segs; base += 64u) {
        uint32_t slice = minu(n_segs - base, 64u);
        wave_lds_sync();
        if (lane < slice) sh.seg[lane] = base == 0u ? first : segments[fill.seg_data + base + lane];
        wave_lds_sync();
        for (uint32_t i = 0; i < slice; i++) {
            Segment sg = sh.seg[i];
            float y = sg.p0y - xy_y;
            float delta_x = sg.p1x - sg.p0x;
            float delta_y = sg.p1y - sg.p0y;
            float y0 = clampf(y, 0.0f, 1.0f);
            float y1 = clampf(y + delta_y, 0.0f, 1.0f);
            float dy = y0 - y1;
            if (dy != 0.0f) {
                float vec_y_recip = 1.0f / delta_y;
                float t0 = (y0 - y) * vec_y_recip;
                float t1 = (y1 - y) * vec_y_recip;
                float startx = sg.p0x - xy_x;
                float x0 = startx + t0 * delta_x;
                float x1 = startx + t1 * delta_x;
                float xmin0 = minf(x0, x1);
                float xmax0 = maxf(x0, x1);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float i_f = (float)k;
                    float xmin = minf(xmin0 - i_f, 1.0f) - 1.0e-6f;
                    float xmax = xmax0 - i_f;
                    float b = minf(xmax, 1.0f);
                    float c = maxf(b, 0.0f);
                    float d = maxf(xmin, 0.0f);
                    float a = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                    area[k] += a * dy;
                }
            }
            float y_edge = signf(delta_x) * clampf(xy_y - sg.y_edge + 1.0f, 0.0f, 1.0f);
#pragma unroll
            for (int k = 0; k < 4; k++) area[k] += y_edge;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float a = area[k];
        if (even_odd) a = fabsf(a - 2.0f * roundf_te(0.5f * a));
        else a = minf(fabsf(a), 1.0f);
        area[k] = a;
    }
}

// ---------------- MSAA (fine.wgsl:146-709) ----------------
// The reference rasterizes one fill at a time: count pixel crossings per segment, prefix-sum, one thread per crossing
// ("item") computes a sample mask and bumps packed winding counters in workgroup memory, then every thread resolves
// its 4 pixels.  On the paris-like scene a tile holds ~13 fills of ~8 segments / ~30 items, so a wave64 runs the
// 275-instruction item pass at <50 % lane use 13 times, each time behind the same chain of dependent latencies
// (segment load -> count -> scan -> search -> LUT load -> LDS atomics -> resolve).  Here the arithmetic of fine.wgsl
// is split into three pure pieces -- ms_item (crossing -> 28-bit record), ms_apply (record -> counter atomics),
// ms_resolve -- and up to MS_BATCH_FILLS consecutive fills of the command list are batched: one segment load,
// one count/scan, one dense item pass writing records to LDS; each FILL command then only replays its records
// (ms_apply) and resolves.  Every integer operation on the counters is the reference's, so coverage is bit-identical.
constexpr uint32_t MS_BATCH_FILLS = 12u;    // fills per batch (their segments must fit one 64-lane load)
static_assert(MS_BATCH_FILLS <= 16u, "the slot search of ms_build_batch covers 16 slots");
static_assert(MS_BATCH_FILLS == sizeof(PixelLds::zero_at) / sizeof(PixelLds::zero_at[0]), "a row of PixelLds::zero_at per staged fill");
static_assert(sizeof(PixelLds) <= sizeof(SegSetupLds), "PixelLds lives in SegSetupLds's storage");
static_assert(sizeof(Segment) * 64u > offsetof(PixelLds, zero_at) && sizeof(Segment) * 64u <= offsetof(PixelLds, zero_at) + 4u * sizeof(PixelLds::zero_at[0]),
              "FineShared::seg ends inside rows 0-3 of PixelLds::zero_at: see the invariant at FineShared (seg is never used while a batch is staged)");
#ifndef VK_FINE_ITEM_CAP
#define VK_FINE_ITEM_CAP 512u
#endif
constexpr uint32_t MS_ITEM_CAP = VK_FINE_ITEM_CAP;      // item records per batch (2 KB of LDS)
constexpr uint32_t REC_PIX_VALID = 1u << 24, REC_IS_DOWN = 1u << 25, REC_IS_BUMP = 1u << 26, REC_DELTA_OK = 1u << 27;

struct alignas(16) FineBatch {
    uint32_t item[MS_ITEM_CAP];
    uint32_t seg_slot[64];                     // scratch of the slot-source scatter (lane numbers of the kept fills by rank)
    uint32_t winding_y[MS_BATCH_FILLS][4];     // per fill, as fine.wgsl's sh_winding_y
    float color[MS_BATCH_FILLS][4];            // per fill of the regular prefix: the CMD_COLOR behind it, unpacked once
};
// What a staged fill's replay needs besides its records lives in two registers of lane `slot` (read with v_readlane, no
// LDS round trip at the head of every fill): its record range [begin, end) and fill rule, packed, and its backdrop.
struct SlotRegs {
    uint32_t pack;      // begin | end << 10 | even_odd << 20
    uint32_t backdrop;
};
constexpr uint32_t SLOT_END_SHIFT = 10u, SLOT_EO_SHIFT = 20u, SLOT_IX_MASK = 0x3ffu;
static_assert(MS_ITEM_CAP <= SLOT_IX_MASK, "record indices are packed in 10 bits");

// fine.wgsl:222-330 computes, for every pixel crossing, the line setup of its segment and then the crossing itself.
// The setup (one IEEE division, the robustness fix-up, the LUT row) depends on the segment alone: ms_setup runs once
// per segment lane, ms_item_su once per crossing.  Same operations on the same values, only hoisted.
struct MsSetup {
    float a, b, mask_row;
    int32_t x0i, y0i;
    uint32_t flags;
    uint32_t edge_masks;  // the samples a segment's first crossing keeps (low half) and its last crossing keeps (high half)
};
constexpr uint32_t SU_IS_DOWN = 1u, SU_POS_SLOPE = 2u, SU_DELTA0 = 4u, SU_BUMP0 = 8u, SU_END_OK = 16u;

template <int AA>
__device__ __forceinline__ MsSetup ms_setup(const Segment &sg, bool even_odd) {
    constexpr uint32_t MASK_WIDTH = AA == 2 ? 64u : 32u, MASK_HEIGHT = AA == 2 ? 64u : 32u;
    // line setup, fine.wgsl:236-261
    const bool is_down = sg.p1y >= sg.p0y;
    const vec2 xy0 = is_down ? v2(sg.p0x, sg.p0y) : v2(sg.p1x, sg.p1y);
    const vec2 xy1 = is_down ? v2(sg.p1x, sg.p1y) : v2(sg.p0x, sg.p0y);
    const float dx = fabsf(xy1.x - xy0.x);
    const float dy = xy1.y - xy0.y;
    const float idxdy = 1.0f / (dx + dy);
    float a = dx * idxdy;
    const bool is_positive_slope = xy1.x >= xy0.x;
    const float x_sign = is_positive_slope ? 1.0f : -1.0f;
    const float xt0 = floorf(xy0.x * x_sign);
    const float c = xy0.x * x_sign - xt0;
    const float y0i = floorf(xy0.y);
    const float ytop = y0i + 1.0f;
    const float b = minf((dy * c + dx * (ytop - xy0.y)) * idxdy, ONE_MINUS_ULP);
    const uint32_t count_x = span(xy0.x, xy1.x) - 1u;
    const uint32_t cnt = count_x + span(xy0.y, xy1.y);
    const float robust_err = floorf(a * ((float)cnt - 1.0f) + b) - (float)count_x;
    if (robust_err != 0.0f) a -= ROBUST_EPSILON * signf(robust_err);
    const float half_height = (float)(MASK_HEIGHT / 2u);
    MsSetup su;
    su.a = a;
    su.b = b;
    su.mask_row = floorf(minf(a * half_height, half_height - 1.0f)) * (float)MASK_WIDTH;
    su.x0i = f2i(xt0 * x_sign + 0.5f * (x_sign - 1.0f));
    su.y0i = f2i(y0i);
    // fine.wgsl:316-325 trims the sample mask of a segment's first crossing above its start point and of its last crossing
    // below its end point; the row either crossing lies in follows from the segment alone (crossing 0 and crossing cnt - 1,
    // the same expressions ms_item_su evaluates for them), so both trims are made here, once per segment
    {
        constexpr uint32_t NSAMP = AA == 2 ? 16u : 8u;
        constexpr uint32_t FULL = AA == 2 ? 0xffffu : 0xffu;
        const float z_first = floorf(a * (float)0u + b);
        const int32_t y_first = su.y0i + (int32_t)0u - f2i(z_first);
        const uint32_t shift0 = f2u(roundf_te((float)NSAMP * (xy0.y - (float)y_first)));
        const uint32_t keep_first = (shift0 < 32u ? (FULL << shift0) : 0u) & FULL;
        const uint32_t last_ix = cnt - 1u;
        const float z_last = floorf(a * (float)last_ix + b);
        const int32_t y_last = su.y0i + (int32_t)last_ix - f2i(z_last);
        const uint32_t shift1 = f2u(roundf_te((float)NSAMP * (xy1.y - (float)y_last)));
        const uint32_t keep_last = ~(shift1 < 32u ? (FULL << shift1) : 0u) & FULL;
        su.edge_masks = keep_first | (keep_last << 16);
    }
    const bool is_delta0 = y0i == xy0.y;
    const bool is_bump0 = even_odd ? (xy0.x == 0.0f) : (xy0.x == 0.0f && y0i != xy0.y);
    su.flags = (is_down ? SU_IS_DOWN : 0u) | (is_positive_slope ? SU_POS_SLOPE : 0u) | (is_delta0 ? SU_DELTA0 : 0u) |
               (is_bump0 ? SU_BUMP0 : 0u) | (xy1.x != 0.0f ? SU_END_OK : 0u);
    return su;
}

// One pixel crossing: which pixel, the 8/16-bit sample mask from the LUT, and the flags the accumulation needs.
template <int AA>
__device__ __forceinline__ uint32_t ms_item_su(const MsSetup &su, uint32_t sub_ix, bool last_pixel, const uint32_t *__restrict__ mask_lut) {
    constexpr bool MSAA16 = AA == 2;
    constexpr uint32_t MASK_WIDTH = MSAA16 ? 64u : 32u, MASK_HEIGHT = MSAA16 ? 64u : 32u;
    constexpr uint32_t FULL = MSAA16 ? 0xffffu : 0xffu;
    const bool is_down = (su.flags & SU_IS_DOWN) != 0u, is_positive_slope = (su.flags & SU_POS_SLOPE) != 0u;
    const float a = su.a, b = su.b;
    const float zf = a * (float)sub_ix + b;
    const float z = floorf(zf);
    const int32_t x = su.x0i + f2i(is_positive_slope ? z : -z);  // x_sign * z
    const int32_t y = su.y0i + (int32_t)sub_ix - f2i(z);
    bool is_delta, is_bump;
    const float zp = floorf(a * (float)(sub_ix - 1u) + b);
    if (sub_ix == 0u) {
        is_delta = (su.flags & SU_DELTA0) != 0u;
        is_bump = (su.flags & SU_BUMP0) != 0u;
    } else {
        is_delta = z == zp;
        is_bump = is_positive_slope && !is_delta;
    }
    const uint32_t pix_ix = (uint32_t)y * TILE_WIDTH + (uint32_t)x;
    const bool delta_ok = (uint32_t)x < TILE_WIDTH - 1u && (uint32_t)y < TILE_HEIGHT && is_delta;
    const uint32_t mask_block = (is_positive_slope ? 1u : 0u) * (MASK_WIDTH * MASK_HEIGHT / 2u);
    const float mask_col = floorf((zf - z) * (float)MASK_WIDTH);
    const uint32_t mask_ix = mask_block + f2u(su.mask_row + mask_col);
    uint32_t mask;
    if (MSAA16) mask = (mask_lut[mask_ix / 2u] >> ((mask_ix % 2u) * 16u)) & 0xffffu;
    else mask = (mask_lut[mask_ix / 4u] >> ((mask_ix % 4u) * 8u)) & 0xffu;
    if (sub_ix == 0u && !is_bump) mask &= su.edge_masks & 0xffffu;
    if (last_pixel && (su.flags & SU_END_OK) != 0u) mask &= su.edge_masks >> 16;
    // pix_ix >= 256 only guards memory: tile-clipped segments never produce it.  Such a record is all zeros (delta_ok implies a
    // pixel inside the tile, so nothing of it would be used): a zero record adds zeros wherever it is applied, which is what lets
    // ms_fill_simple apply records without testing them.
    const uint32_t rec = (pix_ix & 0xffu) | ((mask & FULL) << 8) | REC_PIX_VALID | (is_down ? REC_IS_DOWN : 0u) | (is_bump ? REC_IS_BUMP : 0u) |
                         (delta_ok ? REC_DELTA_OK : 0u);
    return pix_ix < 256u ? rec : 0u;
}

template <int AA>
__device__ __forceinline__ uint32_t ms_item(const Segment &sg, uint32_t sub_ix, bool last_pixel, bool even_odd,
                                            const uint32_t *__restrict__ mask_lut) {
    return ms_item_su<AA>(ms_setup<AA>(sg, even_odd), sub_ix, last_pixel, mask_lut);
}

// The counter updates of one crossing (fine.wgsl:262-270, :331-360).
template <int AA>
__device__ __forceinline__ void ms_apply(uint32_t rec, bool even_odd, uint32_t *winding, uint32_t *sh_samples) {
    constexpr bool MSAA16 = AA == 2;
    constexpr uint32_t SWPP = MSAA16 ? 4u : 2u;
    constexpr uint32_t FULL = MSAA16 ? 0xffffu : 0xffu;
    const uint32_t pix_ix = rec & 0xffu;
    const bool is_down = (rec & REC_IS_DOWN) != 0u, is_bump = (rec & REC_IS_BUMP) != 0u;
    if (rec & REC_DELTA_OK) {
        if (!even_odd) {
            uint32_t delta_pix = pix_ix + 1u;
            uint32_t d = (is_down ? 1u : 0xffffffffu) << ((delta_pix & 3u) << 3);
            atomicAdd(&winding[delta_pix >> 2], d);
        } else {
            atomicXor(&winding[pix_ix >> 4], 2u << (pix_ix & 15u));
        }
    }
    if (!(rec & REC_PIX_VALID)) return;
    uint32_t mask = (rec >> 8) & FULL;
    // sample words are stored transposed: logical word w of pixel p lives at ((p & 3) * SWPP + w) * 64 + (p >> 2),
    // so that the words of a pixel and of its x-neighbours land in different banks
    if (even_odd) {
        if (is_bump) mask ^= FULL;
        atomicXor(&sh_samples[(pix_ix & 3u) * 64u + (pix_ix >> 2)], mask);
        return;
    }
    // fine.wgsl adds, per word of four samples, e (the word's mask bits spread one to a byte) or -e by the crossing's
    // direction, plus or minus 0x01010101 for a bump: +(e - bump) upward, -(e - bump) downward, mod 2^32.
    // (Four mask bits to four bytes: bit k of a nibble times 0x204081 lands on bits k, k + 7, k + 14, k + 21 -- no two
    // terms on one bit, so the product is fine.wgsl's shift-and-xor spread, and bits 0, 8, 16, 24 of it are bits 0 .. 3.)
    // (Round 4 measured the sign in the LDS instruction instead -- ds_sub for the downward lanes, ds_add for the others,
    // each under its exec mask: 8 VALU fewer per fill, twice the LDS instructions, no faster; profiles/r04_ab_s12_*.)
    constexpr uint32_t NW = MSAA16 ? 4u : 2u;
    const uint32_t bump = is_bump ? 0x1010101u : 0u;
    uint32_t *word = &sh_samples[(pix_ix & 3u) * SWPP * 64u + (pix_ix >> 2)];
#pragma unroll
    for (uint32_t w = 0; w < NW; w++) {
        const uint32_t v = ((((rec >> (8u + 4u * w)) & 0xfu) * 0x204081u) & 0x1010101u) - bump;
        atomicAdd(&word[w * 64u], is_down ? 0u - v : v);
    }
}

// Per-segment part of the counting stage (fine.wgsl:186-215): crossing count and the row winding bump.
__device__ __forceinline__ uint32_t ms_segment(const Segment &sg, bool even_odd, uint32_t *winding_y) {
    uint32_t count = 0u;
    float y_edge_f = (float)TILE_HEIGHT;
    uint32_t delta = (sg.p1x <= sg.p0x) ? 1u : 0xffffffffu;
    if (sg.p0x == 0.0f) y_edge_f = sg.p0y;
    else if (sg.p1x == 0.0f) y_edge_f = sg.p1y;
    if (!(sg.p0y == sg.p1y && sg.p0y == floorf(sg.p0y))) count = span(sg.p0x, sg.p1x) + span(sg.p0y, sg.p1y) - 1u;
    uint32_t y_edge = f2u(ceilf(y_edge_f));
    if (y_edge < TILE_HEIGHT) {
        if (!even_odd) atomicAdd(&winding_y[y_edge >> 2], delta << ((y_edge & 3u) << 3));
        else atomicXor(&winding_y[0], 1u << y_edge);
    }
    return count;
}

__device__ __forceinline__ void ms_clear(FineShared &sh, uint32_t *sh_samples, bool even_odd, uint32_t lane, uint32_t swpp) {
    if (!even_odd) {
        sh.winding[lane] = 0x80808080u;
        for (uint32_t i = 0; i < PIXELS_PER_THREAD * swpp; i++) sh_samples[i * 64u + lane] = 0x80808080u;
    } else {
        if (lane < 16u) sh.winding[lane] = 0u;
        for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) sh_samples[i * 64u + lane] = 0u;
    }
}

// Resolve (fine.wgsl:365-466): prefix sums of the packed counters, then per pixel the number of covered samples.
template <int AA>
__device__ __forceinline__ void ms_resolve(FineShared &sh, uint32_t *sh_samples, const uint32_t *winding_y, bool even_odd,
                                           int32_t backdrop, uint32_t lane, float (&area)[4]) {
    constexpr bool MSAA16 = AA == 2;
    constexpr uint32_t SWPP = MSAA16 ? 4u : 2u;
    constexpr uint32_t FULL = MSAA16 ? 0xffffu : 0xffu;
    const uint32_t lx = lane & 3u, ly = lane >> 2;
    if (even_odd) {
        uint32_t scan_x = sh.winding[ly];
        scan_x ^= scan_x << 1; scan_x ^= scan_x << 2; scan_x ^= scan_x << 4; scan_x ^= scan_x << 8;
        uint32_t scan_y = winding_y[0];
        scan_y ^= scan_y << 1; scan_y ^= scan_y << 2; scan_y ^= scan_y << 4; scan_y ^= scan_y << 8;
        uint32_t row_parity = (scan_y >> ly) ^ (uint32_t)backdrop;
#pragma unroll
        for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) {
            uint32_t pix_ix = lane * PIXELS_PER_THREAD + i;
            uint32_t samples = sh_samples[i * 64u + lane];
            uint32_t pix_parity = row_parity ^ (scan_x >> (pix_ix % TILE_WIDTH));
            uint32_t pix_mask = (uint32_t)(-(int32_t)(pix_parity & 1u));
            area[i] = (float)__popc((samples ^ pix_mask) & FULL) * (MSAA16 ? 0.0625f : 0.125f);
        }
        return;
    }
    uint32_t packed_w = sh.winding[lane];
    packed_w += (packed_w - 0x808080u) << 8;
    packed_w += (packed_w - 0x8080u) << 16;
    uint32_t packed_y = winding_y[ly >> 2];
    packed_y += (packed_y - 0x808080u) << 8;
    packed_y += (packed_y - 0x8080u) << 16;
    uint32_t wind_y = (packed_y >> ((ly & 3u) << 3)) - 0x80u;
    // fine.wgsl publishes both prefixes through workgroup memory (sh_winding, sh_winding_y_prefix) and re-reads
    // them after a barrier; one wave does it with shuffles.  Integer adds: the order of the terms is irrelevant.
    const uint32_t prefix_x = ((packed_w >> 24) - 0x80u) * 0x1010101u;
    const uint32_t px1 = row_shr<1>(prefix_x), px2 = row_shr<2>(prefix_x), px3 = row_shr<3>(prefix_x);
    if (lx >= 1u) packed_w += px1;
    if (lx >= 2u) packed_w += px2;
    if (lx >= 3u) packed_w += px3;
    // wind_y of rows 3, 7, 11 (any lane of the row holds it)
    const uint32_t wy3 = lane_value<12>(wind_y), wy7 = lane_value<28>(wind_y), wy11 = lane_value<44>(wind_y);
    if (ly >= 4u) wind_y += wy3;
    if (ly >= 8u) wind_y += wy7;
    if (ly >= 12u) wind_y += wy11;
#pragma unroll
    for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) {
        uint32_t expected_zero = (((packed_w >> (i * 8u)) + wind_y) & 0xffu) - (uint32_t)backdrop;
        if (expected_zero >= 256u) {
            area[i] = 1.0f;
        } else if (!MSAA16) {
            uint32_t samples0 = sh_samples[(i * SWPP + 0u) * 64u + lane];
            uint32_t samples1 = sh_samples[(i * SWPP + 1u) * 64u + lane];
            uint32_t xored0 = (expected_zero * 0x1010101u) ^ samples0;
            uint32_t xored0_2 = xored0 | (xored0 * 2u);
            uint32_t xored1 = (expected_zero * 0x1010101u) ^ samples1;
            uint32_t xored1_2 = xored1 | (xored1 >> 1);
            uint32_t xored2 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
            uint32_t xored4 = xored2 | (xored2 * 4u);
            uint32_t xored8 = xored4 | (xored4 * 16u);
            area[i] = (float)__popc(xored8 & 0xC0C0C0C0u) * 0.125f;
        } else {
            uint32_t samples0 = sh_samples[(i * SWPP + 0u) * 64u + lane];
            uint32_t samples1 = sh_samples[(i * SWPP + 1u) * 64u + lane];
            uint32_t samples2 = sh_samples[(i * SWPP + 2u) * 64u + lane];
            uint32_t samples3 = sh_samples[(i * SWPP + 3u) * 64u + lane];
            uint32_t xored0 = (expected_zero * 0x1010101u) ^ samples0;
            uint32_t xored0_2 = xored0 | (xored0 * 2u);
            uint32_t xored1 = (expected_zero * 0x1010101u) ^ samples1;
            uint32_t xored1_2 = xored1 | (xored1 >> 1);
            uint32_t xored01 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
            uint32_t xored01_4 = xored01 | (xored01 * 4u);
            uint32_t xored2 = (expected_zero * 0x1010101u) ^ samples2;
            uint32_t xored2_2 = xored2 | (xored2 * 2u);
            uint32_t xored3 = (expected_zero * 0x1010101u) ^ samples3;
            uint32_t xored3_2 = xored3 | (xored3 >> 1);
            uint32_t xored23 = (xored2_2 & 0xAAAAAAAAu) | (xored3_2 & 0x55555555u);
            uint32_t xored23_4 = xored23 | (xored23 >> 2);
            uint32_t xored4 = (xored01_4 & 0xCCCCCCCCu) | (xored23_4 & 0x33333333u);
            uint32_t xored8 = xored4 | (xored4 * 16u);
            area[i] = (float)__popc(xored8 & 0xF0F0F0F0u) * 0.0625f;
        }
    }
}

// Index of the segment that owns item i: largest el with count[el - 1] <= i (count = inclusive scan in LDS).
__device__ __forceinline__ uint32_t ms_find_segment(const uint32_t *count, uint32_t n, uint32_t i) {
    uint32_t lo = 0u, hi = n;
    while (hi > lo + 1u) {
        uint32_t mid = (lo + hi) >> 1;
        if (i >= count[mid - 1u]) lo = mid; else hi = mid;
    }
    return lo;
}

// One fill on its own, any number of segments: the reference's loop over batches of 64 segments.  Used for fills that
// do not fit a batch.
template <int AA>
__device__ void fill_path_ms(FineShared &sh, uint32_t *sh_samples, const Segment *__restrict__ segments,
                             const uint32_t *__restrict__ mask_lut, CmdFill fill, uint32_t lane, float (&area)[4]) {
    constexpr uint32_t SWPP = AA == 2 ? 4u : 2u;
    const bool even_odd = (fill.size_and_rule & 1u) != 0u;
    const uint32_t n_segs = fill.size_and_rule >> 1;
    wave_lds_sync();
    if (lane < 4u) sh.winding_y[lane] = even_odd ? 0u : 0x80808080u;
    ms_clear(sh, sh_samples, even_odd, lane, SWPP);
    wave_lds_sync();
    const uint32_t n_batch = (n_segs + 63u) / 64u;
    for (uint32_t batch = 0; batch < n_batch; batch++) {
        const uint32_t slice_size = minu(n_segs - batch * 64u, 64u);
        uint32_t count = 0u;
        if (lane < slice_size) {
            Segment sg = segments[fill.seg_data + batch * 64u + lane];
            sh.seg[lane] = sg;
            count = ms_segment(sg, even_odd, sh.winding_y);
        }
        uint32_t incl = wave_incl_scan_u32(count, (int)lane);
        sh.count[lane] = incl;
        uint32_t total = __shfl(incl, 63);
        wave_lds_sync();
        for (uint32_t i = lane; i < total; i += 64u) {
            const uint32_t el_ix = ms_find_segment(sh.count, slice_size, i);
            const bool last_pixel = i + 1u == sh.count[el_ix];
            const uint32_t sub_ix = i - (el_ix > 0u ? sh.count[el_ix - 1u] : 0u);
            Segment sg = sh.seg[el_ix];
            ms_apply<AA>(ms_item<AA>(sg, sub_ix, last_pixel, even_odd, mask_lut), even_odd, sh.winding, sh_samples);
        }
        wave_lds_sync();
    }
    ms_resolve<AA>(sh, sh_samples, sh.winding_y, even_odd, fill.backdrop, lane, area);
}

// Stage a batch: starting at the FILL command at cmd_ix, collect the following FILL commands visible in the command
// window (any other commands in between are skipped over: coverage does not depend on them), load all their segments
// with one instruction, count, scan, and write one record per crossing.  Returns the number of fills staged (0 when
// the first fill alone does not fit: > 64 segments or > MS_ITEM_CAP crossings).
// `n_fast` / `after_fast`: the batch's REGULAR PREFIX -- the list from cmd_ix on reads FILL, COLOR, FILL, COLOR, ... for the
// first n_fast staged fills (what coarse emits for a run of solid-colour paths): the caller replays those slots in a loop of
// its own, blends bt.color[slot] and goes on at after_fast, without a trip round the interpreter per command.
template <int AA>
__device__ uint32_t ms_build_batch(FineShared &sh, FineBatch &bt, const Segment *__restrict__ segments,
                                   const uint32_t *__restrict__ mask_lut, uint32_t win, uint32_t win_base, uint32_t cmd_ix,
                                   uint32_t lane_in, uint32_t &after_batch, FineProf &pf, uint32_t &n_fast, uint32_t &after_fast,
                                   SlotRegs &slots, uint32_t &simple_mask) {
    const uint32_t lane = opaque(lane_in);  // (the LDS addresses and lane tests below are recomputed per batch, not held)
    n_fast = 0u;
    after_fast = 0u;
    simple_mask = 0u;
    // The scan of the window for the FILLs of the batch, by all lanes at once (a scalar walk, one readlane per word with
    // its hazard slots, cost 600+ issue slots per batch).  Lane i looks at word i as if a command started there:
    // next[i] = i + its size, stopping at END / JUMP / unknown tags and where a FILL's four words would leave the window.
    // Pointer doubling then gives lane k the position of the k-th command from cmd_ix (apply next^(2^b) for every set
    // bit b of k); the chain's fixed point is its stopping position, which is never a FILL that is taken.
    const uint32_t s0 = cmd_ix - win_base;
    uint32_t sz = 0u;
    if (win == CMD_FILL) sz = 4u;
    else if (win == CMD_COLOR || win == CMD_IMAGE) sz = 2u;
    else if (win == CMD_SOLID || win == CMD_BEGIN_CLIP) sz = 1u;
    else if (win == CMD_END_CLIP || win == CMD_LIN_GRAD || win == CMD_RAD_GRAD || win == CMD_SWEEP_GRAD || win == CMD_BLUR_RECT) sz = 3u;
    const bool stop_here = sz == 0u || lane > 60u;  // (the walk went on only while ix + 4 <= window end)
    uint32_t jump = stop_here ? lane : minu(lane + sz, 63u);
    uint32_t pos = s0;
#pragma unroll
    for (uint32_t bit = 0; bit < 6u; bit++) {
        const uint32_t hop = wave_shfl(jump, pos);
        if ((lane >> bit) & 1u) pos = hop;
        jump = wave_shfl(jump, jump);
    }
    // lane k: the k-th command of the list sits at `pos`
    const uint32_t tag_k = wave_shfl(win, pos);
    const uint32_t rule_k = wave_shfl(win, minu(pos + 1u, 63u));
    bool is_fill = tag_k == CMD_FILL && pos <= 60u;
    const uint32_t segs_k = is_fill ? rule_k >> 1 : 0u;
    const uint32_t seg_incl = wave_incl_scan_u32(segs_k, (int)lane);
    // the batch ends in front of the first FILL whose segments no longer fit, and after MS_BATCH_FILLS fills
    unsigned long long fills = __ballot(is_fill);
    const unsigned long long over = __ballot(is_fill && seg_incl > 64u);
    if (over != 0ull) fills &= (1ull << (__ffsll((long long)over) - 1)) - 1ull;
    uint32_t n = (uint32_t)__popcll(fills);
    if (n > MS_BATCH_FILLS) {
        // keep the first MS_BATCH_FILLS set bits
        unsigned long long m = fills;
        for (uint32_t k = 0; k < MS_BATCH_FILLS; k++) m &= m - 1ull;
        fills &= ~m;
        n = MS_BATCH_FILLS;
    }
    // slot j (lane j < n) pulls the parameters of the j-th kept fill: the lane of a kept fill leaves its number at its rank
    // among the kept fills (through bt.seg_slot, not in use yet) -- a scalar walk over the set bits cost twelve hoisted lane
    // masks (24 SGPRs held through the whole kernel) and four operations per bit
    uint32_t src = 0u;
    {
        wave_lds_sync();
        if (((fills >> lane) & 1ull) != 0ull) bt.seg_slot[mask_rank_below(fills, lane)] = lane;
        wave_lds_sync();
        if (lane < n) src = bt.seg_slot[lane];
    }
    const uint32_t src_pos = wave_shfl(pos, src);
    uint32_t my_rule_n = wave_shfl(win, minu(src_pos + 1u, 63u));
    uint32_t my_seg_data = wave_shfl(win, minu(src_pos + 2u, 63u));
    uint32_t my_backdrop = wave_shfl(win, minu(src_pos + 3u, 63u));
    uint32_t my_seg_start = wave_shfl(seg_incl - segs_k, src);
    if (lane >= n) {
        my_rule_n = 0u; my_seg_data = 0u; my_backdrop = 0u; my_seg_start = 0u;
    }
    // regular prefix: command k (lane k) is a kept FILL for even k and a CMD_COLOR inside the window for odd k
    uint32_t pairs;
    {
        const bool ok = (lane & 1u) != 0u ? (tag_k == CMD_COLOR && pos <= 62u) : ((fills >> lane) & 1ull) != 0ull;
        const unsigned long long bad = __ballot(!ok);
        pairs = (bad ? (uint32_t)__ffsll((long long)bad) - 1u : 64u) >> 1;
    }
    const uint32_t color_pos = wave_shfl(pos, minu(2u * lane + 1u, 63u));
    const uint32_t my_color = wave_shfl(win, minu(color_pos + 1u, 63u));
    // segments of the kept fills: the inclusive count at the last kept fill
    uint32_t tot_segs = 0u;
    if (n != 0u) tot_segs = wave_read(seg_incl, (uint32_t)(63 - __clzll((long long)fills)));
    if (n == 0u) return 0u;
    // the word behind the last FILL seen: where the caller points its window prefetch (if fewer fills fit, the prefetch
    // is simply not used)
    after_batch = win_base + wave_read(pos, (uint32_t)(63 - __clzll((long long)fills))) + 4u;
    wave_lds_sync();
    if (lane < n) {
        const bool eo = (my_rule_n & 1u) != 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) bt.winding_y[lane][k] = eo ? 0u : 0x80808080u;
    }
    // which slot does staged segment `lane` belong to: the last slot whose first segment is <= lane (starts are
    // non-decreasing; an empty fill shares its start with its successor, which wins) -- a binary search over the slots'
    // starts held in lanes 0 .. n-1, by shuffles
    uint32_t slot = 0u;
#pragma unroll
    for (uint32_t step = 8u; step >= 1u; step >>= 1) {
        const uint32_t probe = slot + step;
        const uint32_t st = wave_shfl(my_seg_start, minu(probe, 63u));
        if (probe < n && st <= lane) slot = probe;
    }
    const uint32_t seg_data = wave_shfl(my_seg_data, slot);
    const uint32_t seg_start = wave_shfl(my_seg_start, slot);
    const uint32_t rule = wave_shfl(my_rule_n, slot);
    wave_lds_sync();
    pf.mark(FP_BATCH_SCAN);
    uint32_t count = 0u;
    if (lane < tot_segs) {
        Segment sg = segments[seg_data + (lane - seg_start)];
        count = ms_segment(sg, (rule & 1u) != 0u, bt.winding_y[slot]);
        const MsSetup su = ms_setup<AA>(sg, (rule & 1u) != 0u);
        sh.su.a[lane] = su.a; sh.su.b[lane] = su.b; sh.su.edge_masks[lane] = su.edge_masks;
        sh.su.mask_row[lane] = su.mask_row; sh.su.x0i[lane] = su.x0i; sh.su.y0i[lane] = su.y0i; sh.su.flags[lane] = su.flags;
    }
    uint32_t incl = wave_incl_scan_u32(count, (int)lane);
    // item range ends per slot: the inclusive count at the slot's last segment (or the previous end for empty fills)
    uint32_t my_end;
    {
        const uint32_t last_seg = my_seg_start + (my_rule_n >> 1);  // one past the slot's last staged segment (0 for lanes >= n)
        const uint32_t end = wave_shfl(incl, last_seg ? last_seg - 1u : 0u);
        my_end = last_seg ? end : 0u;
    }
    // keep only the fills whose records fit
    const unsigned long long fits = __ballot(lane < n && my_end <= MS_ITEM_CAP);  // (ends are non-decreasing)
    const uint32_t n_fit = fits ? 64u - (uint32_t)__clzll((long long)fits) : 0u;
    if (n_fit == 0u) return 0u;
    const uint32_t total = wave_read(my_end, n_fit - 1u);
    {
        uint32_t my_begin = wave_shfl(my_end, lane - 1u & 63u);  // (only lanes < 12 matter)
        if (lane == 0u) my_begin = 0u;
        // the staged fills ms_fill_simple may take: non-zero rule, 1 .. 64 crossing records
        simple_mask = (uint32_t)__ballot(lane < n_fit && (my_rule_n & 1u) == 0u && my_end - my_begin - 1u < 64u);
        slots.pack = (my_begin & SLOT_IX_MASK) | ((my_end & SLOT_IX_MASK) << SLOT_END_SHIFT) | ((my_rule_n & 1u) << SLOT_EO_SHIFT);
        slots.backdrop = my_backdrop;
        // (unpacked by the slot's lane once, read back with one broadcast load per fill: every lane converting the four
        // channels for itself was 10 VALU per fill)
        if (lane < MS_BATCH_FILLS) {
            const vec4 c = unpack4x8unorm(my_color);
            *reinterpret_cast<float4 *>(&bt.color[lane][0]) = make_float4(c.x, c.y, c.z, c.w);
        }
    }
    n_fast = minu(n_fit, pairs);
    if (n_fast != 0u) after_fast = win_base + wave_read(pos, 2u * n_fast - 1u) + 2u;
    pf.mark(FP_BATCH_SEGS);
    pf.count(FP_N_ITEMS, total);
    // Which segment owns record i?  fine.wgsl searches the scanned counts per record (six dependent LDS reads).  Here every
    // segment with crossings drops a marker (first record index << 6 | segment) at its first record's place in the record
    // array itself; a record's owner is then the largest marker at or before it: one read and a DPP max-scan per 64
    // records, the carry handed from round to round.  A record is its segment's last iff the next place holds a marker (or
    // is the end).  The places are overwritten with the records only after the round has read them (a wave's LDS
    // operations are performed in order).
    for (uint32_t i = lane; i < total; i += 64u) bt.item[i] = 0u;
    wave_lds_sync();
    {
        const uint32_t excl = incl - count;
        if (count != 0u && excl < total) bt.item[excl] = 0x80000000u | (excl << 6) | lane;
    }
    wave_lds_sync();
    uint32_t carry = 0u;
    for (uint32_t base = 0u; base < total; base += 64u) {
        const uint32_t i = base + lane;
        const uint32_t raw = i < total ? bt.item[i] : 0u;
        const uint32_t raw_next = i + 1u < total ? bt.item[i + 1u] : 0x80000000u;
        uint32_t m = wave_incl_scan_max_u32(raw, (int)lane);
        m = maxu(m, carry);
        carry = lane_value<63>(m);
        uint32_t rec = 0u;
        if (i < total) {
            const uint32_t el_ix = m & 63u;
            const uint32_t sub_ix = i - ((m >> 6) & 0xffffu);
            const bool last_pixel = raw_next != 0u;
            MsSetup su;
            su.a = sh.su.a[el_ix]; su.b = sh.su.b[el_ix]; su.edge_masks = sh.su.edge_masks[el_ix];
            su.mask_row = sh.su.mask_row[el_ix]; su.x0i = sh.su.x0i[el_ix]; su.y0i = sh.su.y0i[el_ix]; su.flags = sh.su.flags[el_ix];
            rec = ms_item_su<AA>(su, sub_ix, last_pixel, mask_lut);
        }
        wave_lds_sync();
        if (i < total) bt.item[i] = rec;
    }
    wave_lds_sync();
    // The row part of every staged fill's winding prefix, once per batch instead of once per fill.  fine.wgsl:365-398 gives
    // pixel i of row r the byte ((x prefix byte) + wind_y(r)) & 0xff and compares it -- minus the backdrop -- with 0x80 where
    // no crossing touched the pixel.  wind_y(r) depends on the fill alone: the scanned row counters (same 32-bit operations
    // as ms_resolve, so a counter that leaves its byte carries on exactly as there), summed over the words before r's.  What
    // a fill's replay needs of it is the byte an x prefix must hold for the winding number to be zero at row r:
    // (0x80 + backdrop - wind_y(r)) & 0xff, kept replicated in a word so that one xor tests four pixels.  Lane 4 s + k
    // takes word k of slot s (rows 4k .. 4k+3).  (The setup records are dead by now: their storage is PixelLds's.)
    {
        const uint32_t s_ix = lane >> 2, k = lane & 3u;
        uint32_t y = bt.winding_y[minu(s_ix, MS_BATCH_FILLS - 1u)][k];
        y += (y - 0x808080u) << 8;
        y += (y - 0x8080u) << 16;
        const uint32_t tot = (y >> 24) - 0x80u;  // wind_y of the word's last row: what the words after it add
        const uint32_t t1 = wave_shfl(tot, (lane - 1u) & 63u), t2 = wave_shfl(tot, (lane - 2u) & 63u), t3 = wave_shfl(tot, (lane - 3u) & 63u);
        const uint32_t base = (k >= 1u ? t1 : 0u) + (k >= 2u ? t2 : 0u) + (k >= 3u ? t3 : 0u);
        const uint32_t bd = wave_shfl(my_backdrop, s_ix);
        uint32_t z[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            const uint32_t wind_y = (y >> (8u * j)) - 0x80u + base;
            z[j] = ((0x80u + bd - wind_y) & 0xffu) * 0x1010101u;
        }
        if (s_ix < n_fit) *reinterpret_cast<uint4 *>(&sh.px.zero_at[s_ix][4u * k]) = make_uint4(z[0], z[1], z[2], z[3]);
    }
    wave_lds_sync();
    pf.mark(FP_BATCH_ITEMS);
    return n_fit;
}

// The per-pixel part of the resolve (fine.wgsl:399-466) for one pixel of a non-zero fill: `ez` is the byte a sample
// counter holds where the winding number is zero, s0.. the pixel's packed sample counters.
template <int AA>
__device__ __forceinline__ float ms_pixel_area(uint32_t ez, uint32_t samples0, uint32_t samples1, uint32_t samples2, uint32_t samples3) {
    if (AA != 2) {
        uint32_t xored0 = (ez * 0x1010101u) ^ samples0;
        uint32_t xored0_2 = xored0 | (xored0 * 2u);
        uint32_t xored1 = (ez * 0x1010101u) ^ samples1;
        uint32_t xored1_2 = xored1 | (xored1 >> 1);
        uint32_t xored2 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
        uint32_t xored4 = xored2 | (xored2 * 4u);
        uint32_t xored8 = xored4 | (xored4 * 16u);
        return (float)__popc(xored8 & 0xC0C0C0C0u) * 0.125f;
    }
    uint32_t xored0 = (ez * 0x1010101u) ^ samples0;
    uint32_t xored0_2 = xored0 | (xored0 * 2u);
    uint32_t xored1 = (ez * 0x1010101u) ^ samples1;
    uint32_t xored1_2 = xored1 | (xored1 >> 1);
    uint32_t xored01 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
    uint32_t xored01_4 = xored01 | (xored01 * 4u);
    uint32_t xored2 = (ez * 0x1010101u) ^ samples2;
    uint32_t xored2_2 = xored2 | (xored2 * 2u);
    uint32_t xored3 = (ez * 0x1010101u) ^ samples3;
    uint32_t xored3_2 = xored3 | (xored3 >> 1);
    uint32_t xored23 = (xored2_2 & 0xAAAAAAAAu) | (xored3_2 & 0x55555555u);
    uint32_t xored23_4 = xored23 | (xored23 >> 2);
    uint32_t xored4 = (xored01_4 & 0xCCCCCCCCu) | (xored23_4 & 0x33333333u);
    uint32_t xored8 = xored4 | (xored4 * 16u);
    return (float)__popc(xored8 & 0xF0F0F0F0u) * 0.0625f;
}

// A FILL command whose crossings were staged by ms_build_batch: replay its records and resolve.
//
// Non-zero fills are resolved SPARSELY.  fine.wgsl clears all sample counters, lets the crossings bump some, and has
// every thread read and evaluate the counters of its 4 pixels (MSAA16: 16 LDS words and ~160 VALU per lane and fill).
// A fill of a map-like scene puts ~35 crossings into a tile: >= 85 % of the pixels are never touched, and an untouched
// pixel's counters all hold the cleared value 0x80, so its coverage is 1 if the winding number that reaches it is
// non-zero and 0 otherwise -- no counter has to be read.  So: every lane derives `expected_zero` for its 4 pixels from
// the winding prefix sums as before and takes that 0/1 coverage; ONE LANE PER CROSSING RECORD then evaluates the real
// formula for the record's pixel (its expected_zero comes from the owning lane through LDS), stores the coverage for the
// owner to pick up, and puts the pixel's counters back to the cleared value -- which is what leaves the counters
// clean for the next fill without a clearing pass (`clean` tracks that across fills; the even-odd and the
// one-fill-at-a-time paths clear for themselves and leave the counters dirty).  Two records on one pixel compute and
// store the same value.  Same integer operations on the same counter values as the dense resolve: bit-identical.
// `pre` / `pre_begin`: the records [pre_begin, pre_begin + 64) as the PREVIOUS fill requested them while it worked (one
// LDS round trip less at the head of this fill's chain); ~0 when nothing was requested.  On return they describe the
// request made for the fill that follows.
template <int AA>
__device__ void ms_fill_from_batch(FineShared &sh, FineBatch &bt, uint32_t *sh_samples, uint32_t slot, const SlotRegs &slots, uint32_t lane,
                                   float (&area)[4], bool &clean, uint32_t &pre, uint32_t &pre_begin, FineProf &pf) {
    constexpr uint32_t SWPP = AA == 2 ? 4u : 2u;
    // (uniform values: in scalar registers the rule's branches are real branches, not lane masks)
    const uint32_t pack = (uint32_t)__builtin_amdgcn_readlane((int)slots.pack, (int)slot);
    const int32_t backdrop = __builtin_amdgcn_readlane((int)slots.backdrop, (int)slot);
    const bool even_odd = ((pack >> SLOT_EO_SHIFT) & 1u) != 0u;
    const uint32_t begin = pack & SLOT_IX_MASK, end = (pack >> SLOT_END_SHIFT) & SLOT_IX_MASK;
    const uint32_t had = pre, had_begin = pre_begin;
    pre = bt.item[minu(end + lane, MS_ITEM_CAP - 1u)];  // the next fill's records start where this one's end
    pre_begin = end;
    if (even_odd) {
        wave_lds_sync();
        ms_clear(sh, sh_samples, true, lane, SWPP);
        clean = false;
        wave_lds_sync();
        for (uint32_t i = begin + lane; i < end; i += 64u) ms_apply<AA>(bt.item[i], true, sh.winding, sh_samples);
        wave_lds_sync();
        ms_resolve<AA>(sh, sh_samples, bt.winding_y[slot], true, backdrop, lane, area);
        pf.mark(FP_FILL_EVENODD);
        return;
    }
    wave_lds_sync();
    if (!clean) {
        ms_clear(sh, sh_samples, false, lane, SWPP);
        clean = true;
        wave_lds_sync();
    }
    // (a fill of <= 64 crossings -- nearly all of them -- keeps its record in a register through the three passes)
    const bool one_round = end - begin <= 64u;
    uint32_t rec0 = 0u;
    if (one_round) {
        if (had_begin == begin) rec0 = had;
        else rec0 = bt.item[minu(begin + lane, MS_ITEM_CAP - 1u)];
        if (begin + lane >= end) rec0 = 0u;
        ms_apply<AA>(rec0, false, sh.winding, sh_samples);  // (a zero record does nothing)
    } else {
        for (uint32_t i = begin + lane; i < end; i += 64u) ms_apply<AA>(bt.item[i], false, sh.winding, sh_samples);
    }
    wave_lds_sync();
    pf.mark(FP_FILL_APPLY);
    // x winding prefix sums exactly as ms_resolve (the counters go back to their cleared value as they are read); the row
    // part comes ready-made from ms_build_batch: `zero_at` holds, for this fill and the lane's row, the byte an x prefix
    // holds where the winding number is zero -- expected_zero == 0x80 in fine.wgsl's terms, the value an untouched pixel's
    // sample counters all hold -- in all four bytes of a word: one xor tests the lane's four pixels
    const uint32_t lx = lane & 3u, ly = lane >> 2;
    uint32_t packed_w = sh.winding[lane];
    sh.winding[lane] = 0x80808080u;
    packed_w += (packed_w - 0x808080u) << 8;
    packed_w += (packed_w - 0x8080u) << 16;
    // ((packed_w >> 24) - 0x80) * 0x1010101, mod 2^32: the lane's total, for the lanes to its right in the row of pixels.
    // Masked at the SOURCE (a lane hands its total only as far as its own row of pixels reaches: 3, 2, 1 lanes to the
    // right from columns 0, 1, 2), so that the three shifted adds need no select behind them
    const uint32_t prefix_x = bcast_byte3(packed_w) - 0x80808080u;
    packed_w += row_shr0<1>(lx <= 2u ? prefix_x : 0u);
    packed_w += row_shr0<2>(lx <= 1u ? prefix_x : 0u);
    packed_w += row_shr0<3>(lx == 0u ? prefix_x : 0u);
    // expected_zero = ((x prefix + wind_y) & 0xff) - backdrop is 0x80 iff the x prefix byte equals zero_at -- provided
    // 0x80 + backdrop is a byte at all (a scalar test: beyond it no pixel of the fill has winding number zero)
    const bool zero_possible = (uint32_t)backdrop + 128u < 256u;
    const uint32_t differs = packed_w ^ sh.px.zero_at[slot][ly];
    // untouched counters are 0x80: every sample differs from expected_zero, or none does -- coverage 1 or 0.  Four pixels at
    // once: bit 7 of a byte of `nz` says the byte of `differs` is not zero; shifted down the bytes are 1 or 0 and the
    // byte-to-float conversion makes 1.0f or 0.0f of them
    uint32_t nz = (((differs & 0x7f7f7f7fu) + 0x7f7f7f7fu) | differs) & 0x80808080u;
    if (!zero_possible) nz = 0x80808080u;
    const uint32_t ones = nz >> 7;
#pragma unroll
    for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) area[i] = (float)((ones >> (i * 8u)) & 0xffu);
    if (begin == end) {
        pf.mark(FP_FILL_PREFIX);
        return;
    }
    sh.px.pw[lane] = packed_w;
    *reinterpret_cast<float4 *>(&sh.px.area[lane * 4u]) = make_float4(area[0], area[1], area[2], area[3]);
    wave_lds_sync();
    pf.mark(FP_FILL_PREFIX);
    // expected_zero of pixel p for the lane that holds one of its crossing records: the x prefix byte from the owner's word,
    // the row part back out of zero_at (wind_y = 0x80 + backdrop - zero_at mod 256)
    auto expected_zero_of = [&](uint32_t pix_ix) -> uint32_t {
        const uint32_t xb = sh.px.pw[pix_ix >> 2] >> ((pix_ix & 3u) << 3);
        const uint32_t b = (xb - sh.px.zero_at[slot][pix_ix >> 4] + 0x80u + (uint32_t)backdrop) & 0xffu;
        return b - (uint32_t)backdrop;
    };
    if (one_round) {
        // one record per lane: read the pixel's counters, put them back to the cleared value at once and evaluate.  A wave's
        // LDS instructions are performed in order, so every lane's reads precede any lane's writes (two records on one pixel
        // both see the counters as the crossings left them).
        const bool valid = (rec0 & REC_PIX_VALID) != 0u;
        const uint32_t pix_ix = rec0 & 0xffu;
        const uint32_t so = (pix_ix & 3u) * SWPP * 64u + (pix_ix >> 2);
        uint32_t e = 256u, s0 = 0u, s1 = 0u, s2 = 0u, s3 = 0u;
        if (valid) {
            e = expected_zero_of(pix_ix);
            s0 = sh_samples[so]; s1 = sh_samples[so + 64u];
            if (AA == 2) { s2 = sh_samples[so + 128u]; s3 = sh_samples[so + 192u]; }
        }
        wave_lds_sync();
        if (valid) {
#pragma unroll
            for (uint32_t w = 0; w < SWPP; w++) sh_samples[so + w * 64u] = 0x80808080u;
            if (e < 256u) sh.px.area[pix_ix] = ms_pixel_area<AA>(e, s0, s1, s2, s3);  // (>= 256: coverage 1, as the owner has it)
        }
        wave_lds_sync();
        pf.mark(FP_FILL_SPARSE);
    } else {
        for (uint32_t i0 = begin; i0 < end; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const uint32_t rec = i < end ? bt.item[i] : 0u;
            if (rec & REC_PIX_VALID) {
                const uint32_t pix_ix = rec & 0xffu;
                const uint32_t e = expected_zero_of(pix_ix);
                const uint32_t so = (pix_ix & 3u) * SWPP * 64u + (pix_ix >> 2);
                const uint32_t s0 = sh_samples[so], s1 = sh_samples[so + 64u];
                const uint32_t s2 = AA == 2 ? sh_samples[so + 128u] : 0u, s3 = AA == 2 ? sh_samples[so + 192u] : 0u;
                if (e < 256u) sh.px.area[pix_ix] = ms_pixel_area<AA>(e, s0, s1, s2, s3);
            }
        }
        wave_lds_sync();
        pf.mark(FP_FILL_SPARSE);
        for (uint32_t i0 = begin; i0 < end; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const uint32_t rec = i < end ? bt.item[i] : 0u;
            if (rec & REC_PIX_VALID) {
                const uint32_t pix_ix = rec & 0xffu;
                const uint32_t so = (pix_ix & 3u) * SWPP * 64u + (pix_ix >> 2);
#pragma unroll
                for (uint32_t w = 0; w < SWPP; w++) sh_samples[so + w * 64u] = 0x80808080u;
            }
        }
    }
    const float4 a = *reinterpret_cast<const float4 *>(&sh.px.area[lane * 4u]);
    area[0] = a.x; area[1] = a.y; area[2] = a.z; area[3] = a.w;
    pf.mark(FP_FILL_RESTORE);
}

// ms_fill_from_batch for the fill it is nearly always asked for -- non-zero rule, 1 .. 64 crossing records, the sample counters
// clean -- as straight-line code (round 5).  The general routine decides even-odd / clean / one round / records prefetched /
// no records / which lanes hold a record with scalar branches and exec-mask ladders: 70 scalar instructions per fill around
// ~180 vector ones, and a scalar instruction takes an issue slot like a vector one.  Here a lane without a record
// holds the ZERO record, which adds zeros to the counters and the winding word of a pixel of the lane's own, reads that pixel's
// counters, puts the cleared value back where it already is (or where the pixel's own lane puts it in the same instruction) and
// sends its coverage to a word of its own: the same LDS operations on the same values for every lane that does hold a record, nothing conditional but the
// address of the last store.  Coverage is bit-identical to ms_fill_from_batch's (tests: every MSAA image against the oracle).
template <int AA>
__device__ __forceinline__ void ms_fill_simple(FineShared &sh, FineBatch &bt, uint32_t *sh_samples, uint32_t slot, int32_t backdrop, uint32_t rec, uint32_t lane,
                                               float (&area)[4], FineProf &pf) {
    constexpr bool MSAA16 = AA == 2;
    constexpr uint32_t SWPP = MSAA16 ? 4u : 2u;
    // (`rec`: the fill's record of this lane, zero beyond its range -- fetched a fill ahead by the caller's run loop, round 6)
    wave_lds_sync();
#if VK_FINE_REGS
    // ---- Prototype (round 6): the touched pixels' coverage from their RECORDS, in registers ----
    // A record adds +-(bit - bump) to each sample counter of its pixel: sigma * M per sample with M = mask (no bump) or ~mask
    // (bump) and sigma = -1 iff downward XOR bump.  A pixel with one or two records therefore holds counters 0x80 + C with
    // C in {-2 .. 2} known from the two records alone, and its coverage -- the samples whose counter differs from expected_zero --
    // is 16 - popcount(C == expected_zero - 0x80), a handful of bit operations on the 16-bit planes: no counter is touched.  The
    // lanes of a pixel find each other through LDS (a byte per pixel: who holds a record of it; a word per holder: who else).
    // A fill in which some pixel holds THREE or more records (round joins, 46 % of the road map's fills) takes the counters as
    // before -- decided per wave.
    {
        constexpr uint32_t FULL = MSAA16 ? 0xffffu : 0xffu;
        constexpr uint32_t NS = MSAA16 ? 16u : 8u;
        const bool has = rec != 0u;
        const uint32_t pix = has ? rec & 0xffu : lane * 4u;
        const bool down = (rec & REC_IS_DOWN) != 0u;
        {   // the winding word of the pixel to the right, as below
            const uint32_t delta_pix = pix + 1u;
            uint32_t d = (down ? 1u : 0xffffffffu) << ((delta_pix & 3u) << 3);
            if (!(rec & REC_DELTA_OK)) d = 0u;
            atomicAdd(&sh.winding[(delta_pix >> 2) & 63u], d);
        }
        // (no LDS of its own -- 256 bytes more would cost the kernel its seventeenth wave per CU: `sh.count`, idle while a batch is
        // staged, is the table of a byte per pixel, the lane that holds a record of it; `bt.seg_slot`, ms_build_batch's scratch, the
        // word per holder: the lane with the pixel's second record, ~0 for none)
        uint8_t *holder = reinterpret_cast<uint8_t *>(sh.count);
        uint32_t *second = bt.seg_slot;
        second[lane] = 0xffffffffu;
        if (has) holder[pix] = (uint8_t)lane;
        wave_lds_sync();
        const uint32_t o = holder[pix];
        const bool loser = has && o != lane;
        if (loser) second[o] = lane;
        wave_lds_sync();
        const uint32_t chk = second[o & 63u];
        const bool crowded = FINE_WAVE_ANY(loser && chk != lane);  // some pixel holds three records or more
        if (!crowded) {
            // x winding prefix and the 0 / 1 coverage of the untouched pixels: as below
            const uint32_t lx = lane & 3u, ly = lane >> 2;
            uint32_t packed_w = sh.winding[lane];
            sh.winding[lane] = 0x80808080u;
            packed_w += (packed_w - 0x808080u) << 8;
            packed_w += (packed_w - 0x8080u) << 16;
            const uint32_t prefix_x = bcast_byte3(packed_w) - 0x80808080u;
            packed_w += row_shr0<1>(lx <= 2u ? prefix_x : 0u);
            packed_w += row_shr0<2>(lx <= 1u ? prefix_x : 0u);
            packed_w += row_shr0<3>(lx == 0u ? prefix_x : 0u);
            const bool zero_possible = (uint32_t)backdrop + 128u < 256u;
            const uint32_t differs = packed_w ^ sh.px.zero_at[slot][ly];
            uint32_t nz = (((differs & 0x7f7f7f7fu) + 0x7f7f7f7fu) | differs) & 0x80808080u;
            if (!zero_possible) nz = 0x80808080u;
            const uint32_t ones = nz >> 7;
#pragma unroll
            for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) area[i] = (float)((ones >> (i * 8u)) & 0xffu);
            sh.px.pw[lane] = packed_w;
            *reinterpret_cast<float4 *>(&sh.px.area[lane * 4u]) = make_float4(area[0], area[1], area[2], area[3]);
            // the pixel's other record (its holder's lane number was left with this lane, if there is one)
            const uint32_t partner = second[lane];
            uint32_t rec2 = wave_shfl(rec, partner & 63u);
            if (partner == 0xffffffffu) rec2 = 0u;
            wave_lds_sync();
            pf.mark(FP_FILL_PREFIX);
            const uint32_t xb = sh.px.pw[pix >> 2] >> ((pix & 3u) << 3);
            const uint32_t eb = (xb - sh.px.zero_at[slot][pix >> 4] + 0x80u + (uint32_t)backdrop) & 0xffu;
            const uint32_t e = eb - (uint32_t)backdrop;
            const uint32_t b1 = (uint32_t)((int32_t)(rec << 5) >> 31), b2 = (uint32_t)((int32_t)(rec2 << 5) >> 31);   // REC_IS_BUMP: bit 26
            const uint32_t m1 = ((rec >> 8) ^ b1) & FULL, m2 = ((rec2 >> 8) ^ b2) & FULL;
            const bool neg1 = (((rec >> 25) ^ (rec >> 26)) & 1u) != 0u, neg2 = (((rec2 >> 25) ^ (rec2 >> 26)) & 1u) != 0u;
            const bool same = neg1 == neg2;
            const int32_t dz = (int32_t)e - 0x80;          // the counter value that means "winding number zero", relative to clear
            const int32_t t = neg1 ? -dz : dz;             // ... in units of the first record's sign
            const uint32_t x = m1 ^ m2, y = m1 & m2, u = m1 | m2;
            uint32_t eq = 0u;                              // the samples whose counter equals expected_zero
            if (t == 0) eq = same ? ~u : ~x;
            if (t == 1) eq = same ? x : x & m1;
            if (t == 2) eq = same ? y : 0u;
            if (t == -1) eq = same ? 0u : x & m2;
            const float cov = (float)(NS - (uint32_t)__popc(eq & FULL)) * (1.0f / (float)NS);
            // (e >= 256: coverage 1, as the pixel's lane has it)
            if (has && !loser && e < 256u) sh.px.area[pix] = cov;
            wave_lds_sync();
            pf.mark(FP_FILL_SPARSE);
            const float4 a = *reinterpret_cast<const float4 *>(&sh.px.area[lane * 4u]);
            area[0] = a.x; area[1] = a.y; area[2] = a.z; area[3] = a.w;
            pf.mark(FP_FILL_RESTORE);
            return;
        }
    }
    constexpr bool DELTA_DONE = true;
#else
    constexpr bool DELTA_DONE = false;
#endif
    // ---- the records into the counters (ms_apply, non-zero rule, without its tests) ----
    // (a lane without a record works on a pixel of its own, 4 x lane: thirty idle lanes adding their zeros to ONE word are thirty
    // LDS atomics in a row -- k_fine 150 -> 226 us on the road map when they all took pixel 0, profiles/r05_ab_fine_simple.txt)
    const uint32_t pix_ix = rec != 0u ? rec & 0xffu : lane * 4u;
    const bool is_down = (rec & REC_IS_DOWN) != 0u;
    {
        const uint32_t delta_pix = pix_ix + 1u;
        uint32_t d = (is_down ? 1u : 0xffffffffu) << ((delta_pix & 3u) << 3);
        if (!(rec & REC_DELTA_OK)) d = 0u;
        if (!DELTA_DONE) atomicAdd(&sh.winding[(delta_pix >> 2) & 63u], d);  // (& 63: pixel 255 of a record without a delta)
        const uint32_t bump = (rec & REC_IS_BUMP) != 0u ? 0x1010101u : 0u;
        uint32_t *word = &sh_samples[(pix_ix & 3u) * SWPP * 64u + (pix_ix >> 2)];
#pragma unroll
        for (uint32_t w = 0; w < SWPP; w++) {
            const uint32_t v = ((((rec >> (8u + 4u * w)) & 0xfu) * 0x204081u) & 0x1010101u) - bump;
            atomicAdd(&word[w * 64u], is_down ? 0u - v : v);
        }
    }
    wave_lds_sync();
    pf.mark(FP_FILL_APPLY);
    // ---- x winding prefix, 0 / 1 coverage of the untouched pixels: as ms_fill_from_batch ----
    const uint32_t lx = lane & 3u, ly = lane >> 2;
    uint32_t packed_w = sh.winding[lane];
    sh.winding[lane] = 0x80808080u;
    packed_w += (packed_w - 0x808080u) << 8;
    packed_w += (packed_w - 0x8080u) << 16;
    const uint32_t prefix_x = bcast_byte3(packed_w) - 0x80808080u;
    packed_w += row_shr0<1>(lx <= 2u ? prefix_x : 0u);
    packed_w += row_shr0<2>(lx <= 1u ? prefix_x : 0u);
    packed_w += row_shr0<3>(lx == 0u ? prefix_x : 0u);
    const bool zero_possible = (uint32_t)backdrop + 128u < 256u;
    const uint32_t differs = packed_w ^ sh.px.zero_at[slot][ly];
    uint32_t nz = (((differs & 0x7f7f7f7fu) + 0x7f7f7f7fu) | differs) & 0x80808080u;
    if (!zero_possible) nz = 0x80808080u;
    const uint32_t ones = nz >> 7;
#pragma unroll
    for (uint32_t i = 0; i < PIXELS_PER_THREAD; i++) area[i] = (float)((ones >> (i * 8u)) & 0xffu);
    sh.px.pw[lane] = packed_w;
    *reinterpret_cast<float4 *>(&sh.px.area[lane * 4u]) = make_float4(area[0], area[1], area[2], area[3]);
    wave_lds_sync();
    pf.mark(FP_FILL_PREFIX);
    // ---- a lane per record: its pixel's counters read, put back, evaluated ----
    {
        const uint32_t xb = sh.px.pw[pix_ix >> 2] >> ((pix_ix & 3u) << 3);
        const uint32_t eb = (xb - sh.px.zero_at[slot][pix_ix >> 4] + 0x80u + (uint32_t)backdrop) & 0xffu;
        const uint32_t e = eb - (uint32_t)backdrop;
        const uint32_t so = (pix_ix & 3u) * SWPP * 64u + (pix_ix >> 2);
        const uint32_t s0 = sh_samples[so], s1 = sh_samples[so + 64u];
        const uint32_t s2 = MSAA16 ? sh_samples[so + 128u] : 0u, s3 = MSAA16 ? sh_samples[so + 192u] : 0u;
        wave_lds_sync();
#pragma unroll
        for (uint32_t w = 0; w < SWPP; w++) sh_samples[so + w * 64u] = 0x80808080u;
        // (e >= 256: coverage 1, as the pixel's lane has it; a lane without a record: into sh.count, idle while a batch is staged)
        float *dst = rec != 0u && e < 256u ? &sh.px.area[pix_ix] : reinterpret_cast<float *>(&sh.count[lane]);
        *dst = ms_pixel_area<AA>(e, s0, s1, s2, s3);
    }
    wave_lds_sync();
    pf.mark(FP_FILL_SPARSE);
    const float4 a = *reinterpret_cast<const float4 *>(&sh.px.area[lane * 4u]);
    area[0] = a.x; area[1] = a.y; area[2] = a.z; area[3] = a.w;
    pf.mark(FP_FILL_RESTORE);
}

// ---------------- blend (shared/blend.wgsl) ----------------
struct vec3 {
    float x, y, z;
};
__device__ __forceinline__ vec3 v3(float x, float y, float z) { return vec3{x, y, z}; }
__device__ __forceinline__ float min3(vec3 c) { return minf(c.x, minf(c.y, c.z)); }
__device__ __forceinline__ float max3(vec3 c) { return maxf(c.x, maxf(c.y, c.z)); }
__device__ __forceinline__ float lum(vec3 c) { return c.x * 0.3f + c.y * 0.59f + c.z * 0.11f; }
__device__ __forceinline__ float svg_lum(vec3 c) { return c.x * 0.2125f + c.y * 0.7154f + c.z * 0.0721f; }
__device__ __forceinline__ float sat(vec3 c) { return max3(c) - min3(c); }
__device__ __forceinline__ float screen1(float cb, float cs) { return cb + cs - (cb * cs); }
__device__ float color_dodge(float cb, float cs) {
    if (cb == 0.0f) return 0.0f;
    else if (cs == 1.0f) return 1.0f;
    else return minf(1.0f, cb / (1.0f - cs));
}
__device__ float color_burn(float cb, float cs) {
    if (cb == 1.0f) return 1.0f;
    else if (cs == 0.0f) return 0.0f;
    else return 1.0f - minf(1.0f, (1.0f - cb) / cs);
}
__device__ __forceinline__ float hard_light1(float cb, float cs) { return cs <= 0.5f ? cb * 2.0f * cs : screen1(cb, 2.0f * cs - 1.0f); }
__device__ __forceinline__ float soft_light1(float cb, float cs) {
    float d = cb <= 0.25f ? ((16.0f * cb - 12.0f) * cb + 4.0f) * cb : sqrtf(cb);
    return cs <= 0.5f ? cb - (1.0f - 2.0f * cs) * cb * (1.0f - cb) : cb + (2.0f * cs - 1.0f) * (d - cb);
}
__device__ vec3 clip_color(vec3 c) {
    float l = lum(c);
    float n = min3(c);
    float x = max3(c);
    if (n < 0.0f) c = v3(l + (((c.x - l) * l) / (l - n)), l + (((c.y - l) * l) / (l - n)), l + (((c.z - l) * l) / (l - n)));
    if (x > 1.0f)
        c = v3(l + (((c.x - l) * (1.0f - l)) / (x - l)), l + (((c.y - l) * (1.0f - l)) / (x - l)), l + (((c.z - l) * (1.0f - l)) / (x - l)));
    return c;
}
__device__ vec3 set_lum(vec3 c, float l) {
    float d = l - lum(c);
    return clip_color(v3(c.x + d, c.y + d, c.z + d));
}
__device__ __forceinline__ void set_sat_inner(float &cmin, float &cmid, float &cmax, float s) {
    if (cmax > cmin) {
        cmid = ((cmid - cmin) * s) / (cmax - cmin);
        cmax = s;
    } else {
        cmid = 0.0f;
        cmax = 0.0f;
    }
    cmin = 0.0f;
}
__device__ vec3 set_sat(vec3 c, float s) {
    float r = c.x, g = c.y, b = c.z;
    if (r <= g) {
        if (g <= b) set_sat_inner(r, g, b, s);
        else if (r <= b) set_sat_inner(r, b, g, s);
        else set_sat_inner(b, r, g, s);
    } else {
        if (r <= b) set_sat_inner(g, r, b, s);
        else if (g <= b) set_sat_inner(g, b, r, s);
        else set_sat_inner(b, g, r, s);
    }
    return v3(r, g, b);
}
__device__ vec3 blend_mix(vec3 cb, vec3 cs, uint32_t mode) {
    switch (mode) {
    case 1: return v3(cb.x * cs.x, cb.y * cs.y, cb.z * cs.z);
    case 2: return v3(screen1(cb.x, cs.x), screen1(cb.y, cs.y), screen1(cb.z, cs.z));
    case 3: return v3(hard_light1(cs.x, cb.x), hard_light1(cs.y, cb.y), hard_light1(cs.z, cb.z));
    case 4: return v3(minf(cb.x, cs.x), minf(cb.y, cs.y), minf(cb.z, cs.z));
    case 5: return v3(maxf(cb.x, cs.x), maxf(cb.y, cs.y), maxf(cb.z, cs.z));
    case 6: return v3(color_dodge(cb.x, cs.x), color_dodge(cb.y, cs.y), color_dodge(cb.z, cs.z));
    case 7: return v3(color_burn(cb.x, cs.x), color_burn(cb.y, cs.y), color_burn(cb.z, cs.z));
    case 8: return v3(hard_light1(cb.x, cs.x), hard_light1(cb.y, cs.y), hard_light1(cb.z, cs.z));
    case 9: return v3(soft_light1(cb.x, cs.x), soft_light1(cb.y, cs.y), soft_light1(cb.z, cs.z));
    case 10: return v3(fabsf(cb.x - cs.x), fabsf(cb.y - cs.y), fabsf(cb.z - cs.z));
    case 11: return v3(cb.x + cs.x - 2.0f * cb.x * cs.x, cb.y + cs.y - 2.0f * cb.y * cs.y, cb.z + cs.z - 2.0f * cb.z * cs.z);
    case 12: return set_lum(set_sat(cs, sat(cb)), lum(cb));
    case 13: return set_lum(set_sat(cb, sat(cs)), lum(cb));
    case 14: return set_lum(cs, lum(cb));
    case 15: return set_lum(cb, lum(cs));
    default: return cs;
    }
}
__device__ vec4 blend_compose(vec3 cb, vec3 cs, float ab, float as_, uint32_t mode) {
    float fa = 0.0f, fb = 0.0f;
    switch (mode) {
    case 1: fa = 1.0f; fb = 0.0f; break;
    case 2: fa = 0.0f; fb = 1.0f; break;
    case 3: fa = 1.0f; fb = 1.0f - as_; break;
    case 4: fa = 1.0f - ab; fb = 1.0f; break;
    case 5: fa = ab; fb = 0.0f; break;
    case 6: fa = 0.0f; fb = as_; break;
    case 7: fa = 1.0f - ab; fb = 0.0f; break;
    case 8: fa = 0.0f; fb = 1.0f - as_; break;
    case 9: fa = ab; fb = 1.0f - as_; break;
    case 10: fa = 1.0f - ab; fb = as_; break;
    case 11: fa = 1.0f - ab; fb = 1.0f - as_; break;
    case 12: fa = 1.0f; fb = 1.0f; break;
    case 13:
        return vec4{minf(1.0f, as_ * cs.x + ab * cb.x), minf(1.0f, as_ * cs.y + ab * cb.y), minf(1.0f, as_ * cs.z + ab * cb.z),
                    minf(1.0f, as_ + ab)};
    default: break;
    }
    float as_fa = as_ * fa, ab_fb = ab * fb;
    return vec4{as_fa * cs.x + ab_fb * cb.x, as_fa * cs.y + ab_fb * cb.y, as_fa * cs.z + ab_fb * cb.z, minf(as_fa + ab_fb, 1.0f)};
}
__device__ __forceinline__ vec3 unpremultiply(vec4 c) {
    float inv_alpha = 1.0f / maxf(c.w, 1e-15f);
    return v3(c.x * inv_alpha, c.y * inv_alpha, c.z * inv_alpha);
}
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ vec4 blend_mix_compose(vec4 backdrop, vec4 src, uint32_t mode) {
    const uint32_t BLEND_DEFAULT = (0u << 8) | 3u;
    if ((mode & 0x7fffu) == BLEND_DEFAULT) return backdrop * (1.0f - src.w) + src;
    vec3 cs = unpremultiply(src);
    vec3 cb = unpremultiply(backdrop);
    uint32_t mix_mode = mode >> 8;
    vec3 mixed = blend_mix(cb, cs, mix_mode);
    cs = v3(mixf(cs.x, mixed.x, backdrop.w), mixf(cs.y, mixed.y, backdrop.w), mixf(cs.z, mixed.z, backdrop.w));
    uint32_t compose_mode = mode & 0xffu;
    if (compose_mode == 3u) {
        return vec4{mixf(backdrop.x, cs.x, src.w), mixf(backdrop.y, cs.y, src.w), mixf(backdrop.z, cs.z, src.w),
                    src.w + backdrop.w * (1.0f - src.w)};
    }
    return blend_compose(cb, cs, backdrop.w, src.w, compose_mode);
}

__device__ __forceinline__ float extend_mode_normalized(float t, uint32_t mode) {  // fine.wgsl:863-875
    switch (mode) {
    case 0: return clampf(t, 0.0f, 1.0f);
    case 1: return t - floorf(t);
    default: return fabsf(t - 2.0f * roundf_te(0.5f * t));
    }
}
__device__ __forceinline__ vec4 ramp_load(const uint32_t *__restrict__ ramps, uint32_t n_ramps, int32_t x, uint32_t index) {
    if (!ramps || index >= n_ramps || x < 0 || x >= GRADIENT_WIDTH) return vec4{0.0f, 0.0f, 0.0f, 0.0f};
    return unpack4x8unorm(ramps[index * GRADIENT_WIDTH + (uint32_t)x]);
}
__device__ __forceinline__ void src_over(vec4 &rgba, vec4 fg, float area) {
    vec4 fg_i = fg * area;
    rgba = rgba * (1.0f - fg_i.w) + fg_i;
}

// ---------------- images (fine.wgsl:804-993, :1315-1382) ----------------
// Kept out of line: image and blur commands are rare and must not cost the solid-colour path registers.
struct Atlas {
    const uint32_t *texels;
    uint32_t w, h;
};
__device__ __forceinline__ vec4 atlas_load(Atlas at, float u, float v, uint32_t alpha_type) {
    // textureLoad(image_atlas, vec2<i32>(uv), 0); out-of-range loads return transparent black
    const int32_t x = f2i(truncf(u)), y = f2i(truncf(v));
    if (!at.texels || x < 0 || y < 0 || (uint32_t)x >= at.w || (uint32_t)y >= at.h) return vec4{0.0f, 0.0f, 0.0f, 0.0f};
    vec4 p = unpack4x8unorm(at.texels[(size_t)y * at.w + (uint32_t)x]);
    if (alpha_type == 0u) {  // maybe_premul_alpha
        p.x *= p.w;
        p.y *= p.w;
        p.z *= p.w;
    }
    return p;
}
__device__ __forceinline__ float extend_mode_px(float t, uint32_t mode, float max) {  // fine.wgsl:877-889
    if (mode == 0u) return clampf(t, 0.0f, max);
    return extend_mode_normalized(t / max, mode) * max;
}
__device__ __forceinline__ float single_weight(float t, float a, float b, float c, float d) { return t * (t * (t * d + c) + b) + a; }
__device__ __forceinline__ void cubic_weights(float fr, float (&w)[4]) {  // Mitchell B = C = 1/3, fine.wgsl:893-931
    w[0] = single_weight(fr, (1.0f / 6.0f) / 3.0f, -(3.0f / 6.0f) / 3.0f - 1.0f / 3.0f, (3.0f / 6.0f) / 3.0f + 2.0f * 1.0f / 3.0f,
                         -(1.0f / 6.0f) / 3.0f - 1.0f / 3.0f);
    w[1] = single_weight(fr, 1.0f - (2.0f / 6.0f) / 3.0f, 0.0f, -3.0f + (12.0f / 6.0f) / 3.0f + 1.0f / 3.0f,
                         2.0f - (9.0f / 6.0f) / 3.0f - 1.0f / 3.0f);
    w[2] = single_weight(fr, (1.0f / 6.0f) / 3.0f, (3.0f / 6.0f) / 3.0f + 1.0f / 3.0f, 3.0f - (15.0f / 6.0f) / 3.0f - 2.0f * 1.0f / 3.0f,
                         -2.0f + (9.0f / 6.0f) / 3.0f + 1.0f / 3.0f);
    w[3] = single_weight(fr, 0.0f, 0.0f, -1.0f / 3.0f, (1.0f / 6.0f) / 3.0f + 1.0f / 3.0f);
}
// Premultiplied sample of the image brush at pixel centre (px, py): read_image + the quality switch.
__device__ __attribute__((noinline)) vec4 image_sample(Atlas at, const uint32_t *__restrict__ info, uint32_t io, float px, float py) {
    const float m0 = __uint_as_float(info[io]), m1 = __uint_as_float(info[io + 1u]), m2 = __uint_as_float(info[io + 2u]),
                m3 = __uint_as_float(info[io + 3u]);
    const float xl0 = __uint_as_float(info[io + 4u]), xl1 = __uint_as_float(info[io + 5u]);
    const uint32_t xy = info[io + 6u], width_height = info[io + 7u], sample_alpha = info[io + 8u];
    const uint32_t alpha_type = (sample_alpha >> 14) & 1u, quality = (sample_alpha >> 12) & 3u;
    const uint32_t x_extend = (sample_alpha >> 10) & 3u, y_extend = (sample_alpha >> 8) & 3u;
    const float ox = (float)(xy >> 16), oy = (float)(xy & 0xffffu);
    const float ew = (float)(width_height >> 16), eh = (float)(width_height & 0xffffu);
    const float amx = ox + ew - 1.0f, amy = oy + eh - 1.0f;
    float u = m0 * px + m2 * py + xl0;
    float v = m1 * px + m3 * py + xl1;
    u = extend_mode_px(u, x_extend, ew);
    v = extend_mode_px(v, y_extend, eh);
    if (quality == 0u) {  // IMAGE_QUALITY_LOW: nearest
        u += ox;
        v += oy;
        return atlas_load(at, clampf(u, ox, amx), clampf(v, oy, amy), alpha_type);
    }
    if (quality == 2u) {  // IMAGE_QUALITY_HIGH: bicubic_sample
        u += ox;
        v += oy;
        const float fx = (u + 0.5f) - floorf(u + 0.5f), fy = (v + 0.5f) - floorf(v + 0.5f);
        float wx[4], wy[4];
        cubic_weights(fx, wx);
        cubic_weights(fy, wy);
        vec4 res{0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < 4; j++) {
            const float vv = clampf(v + ((float)j - 1.5f), oy, amy);
            vec4 s0 = atlas_load(at, clampf(u - 1.5f, ox, amx), vv, alpha_type);
            vec4 s1 = atlas_load(at, clampf(u - 0.5f, ox, amx), vv, alpha_type);
            vec4 s2 = atlas_load(at, clampf(u + 0.5f, ox, amx), vv, alpha_type);
            vec4 s3 = atlas_load(at, clampf(u + 1.5f, ox, amx), vv, alpha_type);
            vec4 row = s0 * wx[0] + s1 * wx[1] + s2 * wx[2] + s3 * wx[3];
            res = j == 0 ? row * wy[0] : res + row * wy[j];
        }
        const float a = clampf(res.w, 0.0f, 1.0f);
        return vec4{clampf(res.x, 0.0f, a), clampf(res.y, 0.0f, a), clampf(res.z, 0.0f, a), a};
    }
    // IMAGE_QUALITY_MEDIUM (default): bilinear
    u = u + ox - 0.5f;
    v = v + oy - 0.5f;
    const float uc = clampf(u, ox, amx), vc = clampf(v, oy, amy);
    const float x0 = floorf(uc), y0 = floorf(vc), x1 = ceilf(uc), y1 = ceilf(vc);
    const float frx = u - floorf(u), fry = v - floorf(v);
    const vec4 a = atlas_load(at, x0, y0, alpha_type), b = atlas_load(at, x0, y1, alpha_type);
    const vec4 c = atlas_load(at, x1, y0, alpha_type), d = atlas_load(at, x1, y1, alpha_type);
    return vec4{mixf(mixf(a.x, b.x, fry), mixf(c.x, d.x, fry), frx), mixf(mixf(a.y, b.y, fry), mixf(c.y, d.y, fry), frx),
                mixf(mixf(a.z, b.z, fry), mixf(c.z, d.z, fry), frx), mixf(mixf(a.w, b.w, fry), mixf(c.w, d.w, fry), frx)};
}

// ---------------- blurred rounded rectangle (fine.wgsl:715-726, :1173-1224) ----------------
__device__ __forceinline__ float erf7(float x) {
    const float y = clampf(x * 1.1283791671f, -100.0f, 100.0f);
    const float yy = y * y;
    const float z = y + (0.24295f + (0.03395f + 0.0104f * yy) * yy) * (y * yy);
    return z / sqrtf(1.0f + z * z);
}
__device__ __forceinline__ float hypot_wgsl(float a, float b) { return sqrtf(a * a + b * b); }
// Coverage factor `alpha` of the blurred rounded rect at pixel corner (px, py).
__device__ __attribute__((noinline)) float blur_rect_alpha(const uint32_t *__restrict__ info, uint32_t io, float px, float py) {
    const float m0 = __uint_as_float(info[io]), m1 = __uint_as_float(info[io + 1u]), m2 = __uint_as_float(info[io + 2u]),
                m3 = __uint_as_float(info[io + 3u]);
    const float xl0 = __uint_as_float(info[io + 4u]), xl1 = __uint_as_float(info[io + 5u]);
    const float bw = __uint_as_float(info[io + 6u]), bh = __uint_as_float(info[io + 7u]);
    const float bradius = __uint_as_float(info[io + 8u]), bstd = __uint_as_float(info[io + 9u]);
    const float std_dev = maxf(bstd, 1e-5f);
    const float inv_std_dev = 1.0f / std_dev;
    const float min_edge = minf(bw, bh);
    const float radius_max = 0.5f * min_edge;
    const float r0 = minf(hypot_wgsl(bradius, std_dev * 1.15f), radius_max);
    const float r1 = minf(hypot_wgsl(bradius, std_dev * 2.0f), radius_max);
    const float exponent = 2.0f * r1 / r0;
    const float inv_exponent = 1.0f / exponent;
    const float delta =
        1.25f * std_dev * (exp_cr(-pow_cr(0.5f * inv_std_dev * bw, 2.0f)) - exp_cr(-pow_cr(0.5f * inv_std_dev * bh, 2.0f)));
    const float width = bw + minf(delta, 0.0f);
    const float height = bh - maxf(delta, 0.0f);
    const float scale = 0.5f * erf7(inv_std_dev * 0.5f * (maxf(width, height) - 0.5f * bradius));
    const float x = m0 * px + m2 * py + xl0;
    const float y = m1 * px + m3 * py + xl1;
    const float y0 = fabsf(y) - (height * 0.5f - r1);
    const float y1 = maxf(y0, 0.0f);
    const float x0 = fabsf(x) - (width * 0.5f - r1);
    const float x1 = maxf(x0, 0.0f);
    const float d_pos = pow_cr(pow_cr(x1, exponent) + pow_cr(y1, exponent), inv_exponent);
    const float d_neg = minf(maxf(x0, y0), 0.0f);
    const float d = d_pos + d_neg - r1;
    return scale * (erf7(inv_std_dev * (min_edge + d)) - erf7(inv_std_dev * d));
}

}  // namespace

// Everything except FILL / SOLID / COLOR / JUMP / END: clip layers (blend stack) and, when BRUSHES, the gradient,
// image and blur arms.  Out of line on purpose: inlined into the interpreter loop these arms cost the hot
// FILL+COLOR path ~250 VALU per fill in register copies at the loop's join points (1391 static VALU in the loop,
// PMC: 52 M of 125 M VALU per frame with the rasterizer removed).  The pixel state crosses the call through
// RareState (scratch); the blend stack lives in scratch permanently, only these commands touch it.
struct RareState {
    vec4 rgba[4];
    float area[4];
    uint32_t clip_depth;
    uint32_t cmd_ix;
};
// (round 5: the body is inlined into the BRUSH kernels' interpreter -- k_fine<.., true> -- and called out of line everywhere else.
// Out of line, the pixel state crosses the call through scratch memory, written by the caller and read back by the callee with
// FLAT loads, and returns the same way: two memory round trips per command, ~10 us under load.  A solid-colour scene meets a clip
// command now and then and its hot loop must not carry these arms; a scene with gradients, images or blend layers meets such a
// command ten times a tile: blend_grid 900^2 fine 185 -> 70 us, gradient_extend 32 -> 21, image_sampling 31 -> 24
// (profiles/r05_brush_prof.txt).)
template <bool BRUSHES>
__device__ __forceinline__ void rare_command_body(RareState &st, uint32_t (*blend_stack)[4], uint32_t tag, uint32_t ptcl_size,
                                                       uint32_t blend_size,
                                                       const uint32_t *__restrict__ ptcl, const uint32_t *__restrict__ info,
                                                       uint32_t *blend_spill, uint32_t blend_offset, uint32_t lane, float xy_x,
                                                       float xy_y, const uint32_t *__restrict__ ramps, uint32_t n_ramps,
                                                       const uint32_t *__restrict__ atlas_texels, uint32_t atlas_w, uint32_t atlas_h) {
    vec4 rgba[4] = {st.rgba[0], st.rgba[1], st.rgba[2], st.rgba[3]};
    float area[4] = {st.area[0], st.area[1], st.area[2], st.area[3]};
    uint32_t clip_depth = st.clip_depth;
    uint32_t cmd_ix = st.cmd_ix;
    auto rd = [&](uint32_t ix) -> uint32_t { return ix < ptcl_size ? ptcl[ix] : 0u; };
    if (tag == CMD_BEGIN_CLIP) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t packed = pack4x8unorm(rgba[i]);
            if (clip_depth < BLEND_STACK_SPLIT) {
                blend_stack[clip_depth][i] = packed;
            } else {
                uint32_t ix = blend_offset + (clip_depth - BLEND_STACK_SPLIT) * TILE_WIDTH * TILE_HEIGHT + lane * 4u + (uint32_t)i;
                if (ix < blend_size) blend_spill[ix] = packed;
            }
            rgba[i] = vec4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        clip_depth += 1u;
        cmd_ix += 1u;
    } else if (tag == CMD_END_CLIP) {
        const uint32_t blend = rd(cmd_ix + 1u);
        const float alpha = __uint_as_float(rd(cmd_ix + 2u));
        clip_depth -= 1u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t bg_rgba = 0u;
            if (clip_depth < BLEND_STACK_SPLIT) {
                bg_rgba = blend_stack[clip_depth][i];
            } else {
                uint32_t ix = blend_offset + (clip_depth - BLEND_STACK_SPLIT) * TILE_WIDTH * TILE_HEIGHT + lane * 4u + (uint32_t)i;
                bg_rgba = ix < blend_size ? blend_spill[ix] : 0u;
            }
            const vec4 bg = unpack4x8unorm(bg_rgba);
            const vec4 fg = (rgba[i] * area[i]) * alpha;
            if (blend == LUMINANCE_MASK_LAYER) {
                if (area[i] == 0.0f) {
                    rgba[i] = bg;
                } else {
                    float luminance = clampf(svg_lum(unpremultiply(fg)) * fg.w, 0.0f, 1.0f);
                    rgba[i] = bg * luminance;
                }
            } else {
                rgba[i] = blend_mix_compose(bg, fg, blend);
            }
        }
        cmd_ix += 3u;
    } else if (BRUSHES && tag == CMD_LIN_GRAD) {
        const uint32_t index_mode = rd(cmd_ix + 1u);
        const uint32_t index = index_mode >> 2, extend = index_mode & 3u;
        const uint32_t io = rd(cmd_ix + 2u);
        const float line_x = __uint_as_float(info[io]), line_y = __uint_as_float(info[io + 1u]), line_c = __uint_as_float(info[io + 2u]);
        const float d = line_x * xy_x + line_y * xy_y + line_c;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float my_d = d + line_x * (float)i;
            int32_t x = f2i(roundf_te(extend_mode_normalized(my_d, extend) * (float)(GRADIENT_WIDTH - 1)));
            src_over(rgba[i], ramp_load(ramps, n_ramps, x, index), area[i]);
        }
        cmd_ix += 3u;
    } else if (BRUSHES && tag == CMD_RAD_GRAD) {
        const uint32_t index_mode = rd(cmd_ix + 1u);
        const uint32_t index = index_mode >> 2, extend = index_mode & 3u;
        const uint32_t io = rd(cmd_ix + 2u);
        const float m0 = __uint_as_float(info[io]), m1 = __uint_as_float(info[io + 1u]), m2 = __uint_as_float(info[io + 2u]),
                    m3 = __uint_as_float(info[io + 3u]);
        const float xl0 = __uint_as_float(info[io + 4u]), xl1 = __uint_as_float(info[io + 5u]);
        const float focal_x = __uint_as_float(info[io + 6u]), radius = __uint_as_float(info[io + 7u]);
        const uint32_t flags_kind = info[io + 8u];
        const uint32_t flags = flags_kind >> 3, kind = flags_kind & 7u;
        const bool is_strip = kind == RAD_GRAD_KIND_STRIP, is_circular = kind == RAD_GRAD_KIND_CIRCULAR;
        const bool is_focal_on_circle = kind == RAD_GRAD_KIND_FOCAL_ON_CIRCLE;
        const bool is_swapped = (flags & RAD_GRAD_SWAPPED) != 0u;
        const float r1_recip = is_circular ? 0.0f : 1.0f / radius;
        const float less_scale = (is_swapped || (1.0f - focal_x) < 0.0f) ? -1.0f : 1.0f;
        const float t_sign = signf(1.0f - focal_x);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float mx = xy_x + (float)i, my = xy_y;
            float x = m0 * mx + m2 * my + xl0;
            float y = m1 * mx + m3 * my + xl1;
            float xx = x * x, yy = y * y;
            float t = 0.0f;
            bool is_valid = true;
            if (is_strip) {
                float a = radius - yy;
                t = sqrtf(a) + x;
                is_valid = a >= 0.0f;
            } else if (is_focal_on_circle) {
                t = (xx + yy) / x;
                is_valid = t >= 0.0f && x != 0.0f;
            } else if (radius > 1.0f) {
                t = sqrtf(xx + yy) - x * r1_recip;
            } else {
                float a = xx - yy;
                t = less_scale * sqrtf(a) - x * r1_recip;
                is_valid = a >= 0.0f && t >= 0.0f;
            }
            if (is_valid) {
                t = extend_mode_normalized(focal_x + t_sign * t, extend);
                if (is_swapped) t = 1.0f - t;
                int32_t rx = f2i(roundf_te(t * (float)(GRADIENT_WIDTH - 1)));
                src_over(rgba[i], ramp_load(ramps, n_ramps, rx, index), area[i]);
            }
        }
        cmd_ix += 3u;
    } else if (BRUSHES && tag == CMD_SWEEP_GRAD) {
        const uint32_t index_mode = rd(cmd_ix + 1u);
        const uint32_t index = index_mode >> 2, extend = index_mode & 3u;
        const uint32_t io = rd(cmd_ix + 2u);
        const float m0 = __uint_as_float(info[io]), m1 = __uint_as_float(info[io + 1u]), m2 = __uint_as_float(info[io + 2u]),
                    m3 = __uint_as_float(info[io + 3u]);
        const float xl0 = __uint_as_float(info[io + 4u]), xl1 = __uint_as_float(info[io + 5u]);
        const float t0 = __uint_as_float(info[io + 6u]), t1 = __uint_as_float(info[io + 7u]);
        const float scale = 1.0f / (t1 - t0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float mx = xy_x + (float)i, my = xy_y;
            float x = m0 * mx + m2 * my + xl0;
            float y = m1 * mx + m3 * my + xl1;
            float xabs = fabsf(x), yabs = fabsf(y);
            float slope = minf(xabs, yabs) / maxf(xabs, yabs);
            float s = slope * slope;
            float phi = slope * (0.15912117063999176025390625f +
                                 s * (-5.185396969318389892578125e-2f +
                                      s * (2.476101927459239959716796875e-2f + s * (-7.0547382347285747528076171875e-3f))));
            if (xabs < yabs) phi = 1.0f / 4.0f - phi;
            if (x < 0.0f) phi = 1.0f / 2.0f - phi;
            if (y < 0.0f) phi = 1.0f - phi;
            if (phi != phi) phi = 0.0f;
            phi = (phi - t0) * scale;
            float t = extend_mode_normalized(phi, extend);
            int32_t rx = f2i(roundf_te(t * (float)(GRADIENT_WIDTH - 1)));
            src_over(rgba[i], ramp_load(ramps, n_ramps, rx, index), area[i]);
        }
        cmd_ix += 3u;
    } else if (BRUSHES && tag == CMD_IMAGE) {
        const uint32_t io = rd(cmd_ix + 1u);
        const uint32_t sample_alpha = info[io + 8u];
        const float alpha = (float)(sample_alpha & 0xFFu) / 255.0f;
        const bool bgra = (sample_alpha >> 15) == 1u;
        const Atlas at{atlas_texels, atlas_w, atlas_h};
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            if (area[i] != 0.0f) {
                vec4 fg = image_sample(at, info, io, xy_x + (float)i + 0.5f, xy_y + 0.5f);
                vec4 fg_i = fg * area[i] * alpha;
                if (bgra) fg_i = vec4{fg_i.z, fg_i.y, fg_i.x, fg_i.w};  // pixel_format: .bgra
                rgba[i] = rgba[i] * (1.0f - fg_i.w) + fg_i;
            }
        }
        cmd_ix += 2u;
    } else if (BRUSHES && tag == CMD_BLUR_RECT) {
        const uint32_t io = rd(cmd_ix + 1u);
        const vec4 blur_rgba = unpack4x8unorm(rd(cmd_ix + 2u));
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            const float alpha = blur_rect_alpha(info, io, xy_x + (float)i, xy_y);
            src_over(rgba[i], blur_rgba * alpha, area[i]);
        }
        cmd_ix += 3u;
    } else if (tag == CMD_LIN_GRAD || tag == CMD_RAD_GRAD || tag == CMD_SWEEP_GRAD || tag == CMD_BLUR_RECT) {
        cmd_ix += 3u;  // !BRUSHES: unreachable by construction; keeps the stream in step if the contract is broken
    } else if (tag == CMD_IMAGE) {
        cmd_ix += 2u;
    } else {
        cmd_ix += 1u;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        st.rgba[i] = rgba[i];
        st.area[i] = area[i];
    }
    st.clip_depth = clip_depth;
    st.cmd_ix = cmd_ix;
}
// (WAVES: a copy per form of the kernel -- an out-of-line function is compiled for the tightest register budget among its callers,
// and the five-wave form's 96 VGPRs made the four-wave form's compositing pass spill: k_fine alone 145 -> 184 us when they shared it)
template <bool BRUSHES, int WAVES>
__device__ __attribute__((noinline)) void rare_command(RareState &st, uint32_t (*blend_stack)[4], uint32_t tag, uint32_t ptcl_size,
                                                       uint32_t blend_size,
                                                       const uint32_t *__restrict__ ptcl, const uint32_t *__restrict__ info,
                                                       uint32_t *blend_spill, uint32_t blend_offset, uint32_t lane, float xy_x,
                                                       float xy_y, const uint32_t *__restrict__ ramps, uint32_t n_ramps,
                                                       const uint32_t *__restrict__ atlas_texels, uint32_t atlas_w, uint32_t atlas_h) {
    rare_command_body<BRUSHES>(st, blend_stack, tag, ptcl_size, blend_size, ptcl, info, blend_spill, blend_offset, lane, xy_x, xy_y, ramps, n_ramps, atlas_texels,
                               atlas_w, atlas_h);
}


// The compositing pass of a sliced tile (k_fine, FINE_SLICE_FILLS): the tile's whole list once more, every FILL's
// coverage read back from the coverage scratch (a byte per pixel, the number of covered samples).  Out of line, called
// once per sliced tile by the wave that finished the tile's last slice -- and the only thing left of the launch's longest
// tiles by then, so what counts is instructions per FILL: a wave pays ~4 ns per instruction whatever it is.
//  * Runs of FILL, COLOR pairs (what a run of solid-colour paths is) are found with ONE parallel scan per 64-word window
//    (next-command pointers doubled over the lanes, as ms_build_batch does); lane j unpacks the colour of pair j once, and
//    the loop over the pairs is a coverage read, four v_readlane and the blend: ~45 instructions per FILL against ~260
//    through a scalar interpreter that decodes every word with readlane.
//  * The window behind the run is requested before the run is composited.
//  * Coverage comes in chunks of COV_CHUNK FILLs: while one chunk is composited out of LDS (the wave's sample counters
//    are idle in this pass), the loads of the next are in flight into registers nothing touches until the hand-over.
//    (Agent-scope loads: they see what the slices' waves wrote through.  One load a fill ahead made the pass as long as
//    its fills' memory round trips; a register ring rotated per fill waits for every load it moves; through a generic
//    pointer the loads are FLAT and any wait for one waits for all.)
// Everything that is not such a pair goes through the one-command-at-a-time arm below, with rare_command as in k_fine.
struct BlendPassOut {
    vec4 rgba[4];
};
template <int AA, bool BRUSHES, int WAVES>
__device__ __attribute__((noinline)) void blend_pass(BlendPassOut &out, uint32_t tile_ix, uint32_t base_color_u, uint32_t ptcl_size, uint32_t blend_size,
                                                     const uint32_t *__restrict__ ptcl, const uint32_t *__restrict__ info, uint32_t *blend_spill,
                                                     const uint32_t *cov_tile, uint32_t *lds_chunk, uint32_t fill_room, uint32_t lane, float xy_x, float xy_y,
                                                     const uint32_t *__restrict__ ramps, uint32_t n_ramps, const uint32_t *__restrict__ atlas_texels,
                                                     uint32_t atlas_w, uint32_t atlas_h) {
    constexpr float COV_SCALE = AA == 2 ? 0.0625f : 0.125f;
    constexpr uint32_t COV_CHUNK = AA == 2 ? 16u : 8u;  // x 64 words: the size of sh_samples
    vec4 rgba[4];
    float area[4];
    {
        const vec4 base_color = unpack4x8unorm(base_color_u);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            rgba[i] = base_color;
            area[i] = 0.0f;
        }
    }
    uint32_t blend_stack[BLEND_STACK_SPLIT][4];
    uint32_t clip_depth = 0u;
    const VK_GLOBAL uint32_t *ptcl_g = as_global(ptcl);
    const VK_GLOBAL uint32_t *cov_g = as_global(cov_tile);
    auto window_at = [&](uint32_t base) -> uint32_t {
        const uint32_t a = base + lane;
        return a < ptcl_size ? ptcl_g[a] : 0u;
    };
    uint32_t cmd_ix = tile_ix * PTCL_INITIAL_ALLOC;
    uint32_t win_base = cmd_ix, win = window_at(cmd_ix);
    uint32_t pf_base = 0xffffffffu, pf_win = 0u;  // the window requested ahead
    const uint32_t blend_offset = wave_read(win, 0u);
    cmd_ix += 1u;
#ifndef VELLO_SIMT_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
    uint32_t fill_no = 0u;
    uint32_t next[COV_CHUNK];
    auto request_chunk = [&](uint32_t first_fill) {
#pragma unroll
        for (uint32_t j = 0; j < COV_CHUNK; j++)  // (the scratch has a fill of slack behind the last)
            next[j] = __hip_atomic_load(&cov_g[minu(first_fill + j, fill_room) * 64u + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    request_chunk(0u);
    // coverage of the FILL fill_no -> area
    auto fill_area = [&]() {
        if (fill_no % COV_CHUNK == 0u) {  // hand the chunk over and request the one behind it
            wave_lds_sync();
#pragma unroll
            for (uint32_t j = 0; j < COV_CHUNK; j++) lds_chunk[j * 64u + lane] = next[j];
            request_chunk(fill_no + COV_CHUNK);
            wave_lds_sync();
        }
        const uint32_t b = lds_chunk[(fill_no % COV_CHUNK) * 64u + lane];
        fill_no += 1u;
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) area[i] = (float)((b >> (8u * i)) & 0xffu) * COV_SCALE;
    };
    for (;;) {
        // a window that starts at the command
        if (cmd_ix != win_base) {
            if (cmd_ix == pf_base) win = pf_win;
            else win = window_at(cmd_ix);
            win_base = cmd_ix;
            pf_base = 0xffffffffu;
        }
        // FILL, COLOR pairs from here on: lane i takes word i for the start of a command, next[i] = i + its size (a fixed
        // point where the command is not one of the two or would leave the window); doubling gives lane k the k-th command
        uint32_t pairs;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f, ca = 0.0f;
        uint32_t after = cmd_ix;
        {
            uint32_t sz = 0u;
            if (win == CMD_FILL && lane <= 60u) sz = 4u;
            else if (win == CMD_COLOR && lane <= 62u) sz = 2u;
            uint32_t jump = sz == 0u ? lane : minu(lane + sz, 63u);
            uint32_t pos = 0u;
#pragma unroll
            for (uint32_t bit = 0; bit < 6u; bit++) {
                const uint32_t hop = wave_shfl(jump, pos);
                if ((lane >> bit) & 1u) pos = hop;
                jump = wave_shfl(jump, jump);
            }
            const uint32_t tag_k = wave_shfl(win, pos);
            const bool ok = (lane & 1u) != 0u ? (tag_k == CMD_COLOR && pos <= 62u) : (tag_k == CMD_FILL && pos <= 60u);
            const unsigned long long bad = __ballot(!ok);
            pairs = (bad ? (uint32_t)__ffsll((long long)bad) - 1u : 64u) >> 1;
            // (two equal positions in a row -- the chain stuck at a word that is neither -- fail the alternation by themselves:
            // a FILL cannot sit where a COLOR is wanted)
            if (pairs != 0u) {
                const uint32_t color_pos = wave_shfl(pos, minu(2u * lane + 1u, 63u));
                const vec4 c = unpack4x8unorm(wave_shfl(win, minu(color_pos + 1u, 63u)));
                cr = c.x; cg = c.y; cb = c.z; ca = c.w;
                after = win_base + wave_read(pos, 2u * pairs - 1u) + 2u;
                pf_base = after;
                pf_win = window_at(after);  // arrives while the run is composited
            }
        }
        if (pairs != 0u) {
            for (uint32_t j = 0; j < pairs; j++) {
                fill_area();
                vec4 fg;
                fg.x = __uint_as_float(wave_read(__float_as_uint(cr), j));
                fg.y = __uint_as_float(wave_read(__float_as_uint(cg), j));
                fg.z = __uint_as_float(wave_read(__float_as_uint(cb), j));
                fg.w = __uint_as_float(wave_read(__float_as_uint(ca), j));
#pragma unroll
                for (int i = 0; i < 4; i++) src_over(rgba[i], fg, area[i]);
            }
            cmd_ix = after;
            continue;
        }
        // one command, as k_fine's interpreter takes it (the window starts at it: its words are lanes 0 ..)
        const uint32_t tag = wave_read(win, 0u);
        if (tag == CMD_END) break;
        if (tag == CMD_FILL) {
            fill_area();
            cmd_ix += 4u;
        } else if (tag == CMD_SOLID) {
#pragma unroll
            for (int i = 0; i < 4; i++) area[i] = 1.0f;
            cmd_ix += 1u;
        } else if (tag == CMD_COLOR) {
            const vec4 fg = unpack4x8unorm(wave_read(win, 1u));
#pragma unroll
            for (int i = 0; i < 4; i++) src_over(rgba[i], fg, area[i]);
            cmd_ix += 2u;
        } else if (tag == CMD_JUMP) {
            cmd_ix = wave_read(win, 1u);
        } else {
            RareState st;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                st.rgba[i] = rgba[i];
                st.area[i] = area[i];
            }
            st.clip_depth = clip_depth;
            st.cmd_ix = cmd_ix;
            rare_command<BRUSHES, WAVES>(st, blend_stack, tag, ptcl_size, blend_size, ptcl, info, blend_spill, blend_offset, lane, xy_x, xy_y, ramps, n_ramps,
                                  atlas_texels, atlas_w, atlas_h);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                rgba[i] = st.rgba[i];
                area[i] = st.area[i];
            }
            clip_depth = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.clip_depth);
            cmd_ix = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.cmd_ix);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out.rgba[i] = rgba[i];
}

// BRUSHES = false is the specialisation for scenes whose draw tags are only COLOR / BEGIN_CLIP / END_CLIP (decided
// on the host when the scene is uploaded): coarse can then never emit a gradient, image or blur command, and the
// solid-colour interpreter does not pay their registers.
// WAVES: the waves per SIMD the register allocation is made for.  4 = 128 VGPRs: four waves a SIMD are the whole register file,
// and while k_fine fills the chip nothing of another frame runs beside it.  With frames in flight the solid-colour MSAA kernels are
// built for 5 (96 VGPRs, 17 spilled; the LDS still admits 17 waves per CU): k_fine alone is 4 % slower (151 for 145 us on d2) and
// the frames/s with four in flight +1.9 % (d2), +1.1 % (mmark-50k) -- the other frames' kernels run in the registers it leaves free
// (profiles/r06_ab_fine_vgpr_cap.txt).  One frame at a time keeps 4.
#ifndef VK_FINE_WAVES_IN_FLIGHT
#define VK_FINE_WAVES_IN_FLIGHT 5
#endif
template <int AA, bool BRUSHES, int WAVES = 4>
__global__ void __launch_bounds__(64, BRUSHES ? 3 : WAVES) k_fine(Config cfg, const Segment *__restrict__ segments, const uint32_t *__restrict__ ptcl,
                                             const uint32_t *__restrict__ info, uint32_t *blend_spill, uint8_t *__restrict__ output,
                                             uint32_t out_stride, const uint32_t *__restrict__ ramps, uint32_t n_ramps,
                                             const uint32_t *__restrict__ mask_lut, const uint32_t *__restrict__ atlas_texels,
                                             uint32_t atlas_w, uint32_t atlas_h, const uint32_t *__restrict__ work_count,
                                             const uint32_t *__restrict__ tile_order, const SliceItem *__restrict__ slice_items,
                                             uint32_t *slice_counters, uint32_t *cov, uint32_t slice_blocks, uint32_t slice_fills) {
    __shared__ FineShared sh;
    __shared__ uint32_t sh_samples[AA == 2 ? 1024 : (AA == 1 ? 512 : 1)];
    __shared__ FineBatch bt;
    if (ptcl[0] == ~0u) return;  // fine.wgsl:1070-1074
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 3u, ly = lane >> 2;
    // Workgroups are dispatched in index order: first the slices of the long tiles (MSAA modes, see FINE_SLICE_FILLS), then
    // index -> tile through coarse's buckets of command-list length, longest
    // lists first, so that the tile that takes longest starts first instead of wherever row-major order puts it.
    const uint32_t n_tiles = cfg.width_in_tiles * cfg.height_in_tiles;
    uint32_t tile_ix = blockIdx.x;
    // A slice's wave runs the interpreter in MODE_COV: it skips to its first FILL, computes the coverage of its fills as
    // any tile's wave does and writes it to the coverage scratch (a byte per pixel: the number of covered samples),
    // ignoring everything else in the list.  The wave that finishes a tile's LAST slice runs the interpreter once more in
    // MODE_BLEND: the whole list, every FILL's coverage read back from the scratch.  Same integer coverage, same f32
    // compositing in the same order as the unsliced tile: bit-identical pixels.
    constexpr uint32_t MODE_NORMAL = 0u, MODE_COV = 1u;
    uint32_t mode = MODE_NORMAL, fill_lo = 0u, fill_hi = 0xffffffffu, fill_room = 0u, cov_base = 0u, slice_n = 0u, first_item = 0u;
    if (AA != 0 && blockIdx.x < slice_blocks) {
        if (blockIdx.x >= minu(work_count[FINE_WORK_BUCKETS], slice_blocks)) return;  // Control::slice_items
        const SliceItem it = slice_items[blockIdx.x];
        if (it.tile_ix >= n_tiles) return;  // a hole
        tile_ix = it.tile_ix;
        const uint32_t k = it.k_and_n & 0xffffu;
        slice_n = it.k_and_n >> 16;
        mode = MODE_COV;
        fill_lo = k * slice_fills;
        fill_hi = k + 1u == slice_n ? 0xffffffffu : (k + 1u) * slice_fills;
        fill_room = slice_n * slice_fills;  // fills the tile's scratch has room for
        cov_base = it.cov_base;
        first_item = it.first_item;
    } else {
        const uint32_t rest = blockIdx.x - (AA != 0 ? slice_blocks : 0u);
        // which bucket, which place in it: lane b holds the count of bucket 31 - b (longest lists first), a wave scan gives
        // the buckets' ends, the first end beyond `rest` is the bucket (a scalar walk over the 32 counters was 450 SALU
        // instructions at the head of every tile's wave)
        static_assert(FINE_WORK_BUCKETS <= 64u, "a lane per bucket");
        const uint32_t cnt = lane < FINE_WORK_BUCKETS ? minu(work_count[FINE_WORK_BUCKETS - 1u - lane], n_tiles) : 0u;
        const uint32_t incl = wave_incl_scan_u32(cnt, (int)lane);
        const unsigned long long beyond = __ballot(incl > rest);
        if (beyond == 0ull) return;
        const uint32_t bl = (uint32_t)__ffsll((long long)beyond) - 1u;  // (< FINE_WORK_BUCKETS: lanes beyond hold the total)
        const uint32_t slot = (FINE_WORK_BUCKETS - 1u - bl) * n_tiles + (rest - wave_read(incl - cnt, bl));
        tile_ix = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_order[slot]);
        if (tile_ix >= n_tiles) return;  // (coarse registers every tile exactly once)
    }
    const uint32_t tile_x = tile_ix % cfg.width_in_tiles, tile_y = tile_ix / cfg.width_in_tiles;
    const float xy_x = (float)(tile_x * TILE_WIDTH + lx * PIXELS_PER_THREAD);
    const float xy_y = (float)(tile_y * TILE_HEIGHT + ly);
    vec4 rgba[4];
    uint32_t blend_stack[BLEND_STACK_SPLIT][4];
    uint32_t clip_depth;
    float area[4];
    uint32_t cmd_ix;
    // The command stream is read 64 words at a time by the whole wave (one 256-B transaction; the tile's
    // initial PTCL block is exactly one window) and decoded with readlane, instead of a dependent scalar
    // load per word.  The segments of the NEXT fill are requested while the current one is rasterized.
    uint32_t win_base;
    uint32_t win;
    auto rd = [&](uint32_t ix) -> uint32_t {
        return (uint32_t)__builtin_amdgcn_readlane((int)win, (int)__builtin_amdgcn_readfirstlane((int)(ix - win_base)));
    };
    auto ensure = [&](uint32_t ix, uint32_t n_words) {
        if (ix + n_words > win_base + 64u) {
            win_base = ix;
            uint32_t a = win_base + lane;
            win = a < cfg.ptcl_size ? ptcl[a] : 0u;
        }
    };
    FineProf prof;
    prof.start();
    FineTimeline tl;
    tl.start();
    uint32_t blend_offset;
    constexpr float COV_SCALE = AA == 2 ? 0.0625f : 0.125f;  // coverage = covered samples / samples
    uint32_t fill_no = 0u;   // sliced tiles: FILLs of the list passed so far
    auto start_list = [&]() {
        const vec4 base_color = unpack4x8unorm(cfg.base_color);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            rgba[i] = base_color;
            area[i] = 0.0f;
        }
        clip_depth = 0u;
        cmd_ix = tile_ix * PTCL_INITIAL_ALLOC;
        win_base = cmd_ix;
        win = ptcl[win_base + lane];
        blend_offset = rd(cmd_ix);
        cmd_ix += 1u;
        fill_no = 0u;
    };
    // a slice's product: covered samples per pixel, a byte each (coverage is k / 8 or k / 16 exactly), for the FILL fill_no
    auto store_cov = [&]() {
        uint32_t b = 0u;
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) b |= (uint32_t)(area[i] * (1.0f / COV_SCALE)) << (8u * i);
        // (an agent-scope relaxed store: written through to where every XCD sees it, no release fence later)
        if (fill_no < fill_room) __hip_atomic_store(&cov[cov_base + fill_no * 64u + lane], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fill_no += 1u;
    };
    start_list();
    prof.count(FP_N_WORDS, rd(cmd_ix + PTCL_INITIAL_ALLOC - 2u));  // the tile's list length, as coarse left it
#ifndef VELLO_SIMT_EMU
    // A long list is the launch's critical path: its wave issues ahead of the waves it shares the SIMD with.
    if (rd(cmd_ix + PTCL_INITIAL_ALLOC - 2u) >= FINE_HEAVY_WORDS) __builtin_amdgcn_s_setprio(3);
#endif
    Segment pre;
    pre.p0x = 0.0f; pre.p0y = 0.0f; pre.p1x = 0.0f; pre.p1y = 0.0f; pre.y_edge = 0.0f; pre.pad = 0u;
    uint32_t pre_seg_data = ~0u;
    uint32_t batch_n = 0u, batch_pos = 0u;  // MSAA: fills staged by ms_build_batch / already consumed
    bool samples_clean = false;             // MSAA: the sample counters hold their cleared (non-zero rule) value
    uint32_t pf_win = 0u, pf_base = 0xffffffffu;  // MSAA: the command window requested ahead for the next batch
    uint32_t fast_left = 0u, fast_after = 0u;     // MSAA: fills left in the batch's regular prefix / where the list goes on behind it
    SlotRegs slots{0u, 0u};                       // MSAA: lane k holds the parameters of the batch's slot k
    uint32_t simple_mask = 0u;                    // MSAA: bit k: slot k is a fill ms_fill_simple may take
    uint32_t rec_pre = 0u, rec_pre_begin = ~0u;   // MSAA: the records requested ahead for the next fill
    for (;;) {
        ensure(cmd_ix, 4u);
        const uint32_t tag = rd(cmd_ix);
        if (tag == CMD_END) break;
        if (tag == CMD_FILL) {
            if constexpr (AA == 0) {
                CmdFill fill;
                fill.size_and_rule = rd(cmd_ix + 1u);
                fill.seg_data = rd(cmd_ix + 2u);
                fill.backdrop = (int32_t)rd(cmd_ix + 3u);
                Segment first = pre;
                if (pre_seg_data != fill.seg_data) {
                    if (lane < minu(fill.size_and_rule >> 1, 64u)) first = segments[fill.seg_data + lane];
                }
                // look ahead: [COLOR] FILL -> prefetch its first segment batch into registers
                pre_seg_data = ~0u;
                uint32_t nx = cmd_ix + 4u;
                if (nx + 2u <= win_base + 64u && rd(nx) == CMD_COLOR) nx += 2u;
                if (nx + 4u <= win_base + 64u && rd(nx) == CMD_FILL) {
                    uint32_t n2 = rd(nx + 1u) >> 1;
                    pre_seg_data = rd(nx + 2u);
                    if (lane < minu(n2, 64u)) pre = segments[pre_seg_data + lane];
                }
                fill_path_area(sh, segments, fill, lane, area, first);
            } else {
                if (mode == MODE_COV && fill_no < fill_lo) {
                    fill_no += 1u;  // in front of the slice
                    cmd_ix += 4u;
                    continue;
                } else {
                if (batch_pos == batch_n) {
                    // the scan wants to see as far ahead as possible: a window that starts at (or a draw command in front
                    // of) this command -- the one requested when the previous batch was staged, if the list went on there
                    if (cmd_ix != win_base) {
                        if (pf_base != 0xffffffffu && cmd_ix >= pf_base && cmd_ix - pf_base <= 8u) {
                            win = pf_win;
                            win_base = pf_base;
                        } else {
                            win_base = cmd_ix;
                            uint32_t a = win_base + lane;
                            win = a < cfg.ptcl_size ? ptcl[a] : 0u;
                        }
                    }
                    uint32_t after_batch = 0xffffffffu;
                    prof.mark(FP_INTERP);
                    prof.count(FP_N_BATCHES, 1u);
                    uint32_t n_fast = 0u, after_fast = 0u, simple = 0u;
                    batch_n = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)ms_build_batch<AA>(sh, bt, segments, mask_lut, win, win_base, cmd_ix, lane, after_batch, prof, n_fast, after_fast, slots, simple));
                    simple_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)simple);
                    batch_pos = 0u;
                    rec_pre_begin = ~0u;
                    pf_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)after_batch);
                    if (pf_base != 0xffffffffu) {
                        const uint32_t a = pf_base + lane;
                        pf_win = a < cfg.ptcl_size ? ptcl[a] : 0u;  // arrives while the batch's fills are replayed
                    }
                    fast_left = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_fast);
                    if (mode == MODE_COV) fast_left = minu(fast_left, fill_hi - fill_no);  // (the slice ends inside the batch)
                    fast_after = (uint32_t)__builtin_amdgcn_readfirstlane((int)after_fast);
                }
                prof.mark(FP_INTERP);
                if (batch_n != 0u) {
                    // The regular prefix of the batch (FILL, COLOR, FILL, COLOR, ...) is replayed and blended in this loop, any
                    // other fill leaves it after one turn and goes on through the interpreter.
                    // Round 6: consecutive SIMPLE fills of the prefix (nearly all of them: non-zero rule, 1 .. 64 records, the
                    // counters clean) are a run with a loop of its own -- a fill there is ms_fill_simple's straight-line code, its
                    // records fetched a fill ahead (a fill's records begin where its predecessor's end), the blend, and a dozen
                    // scalar instructions; the general loop around it spent 38 on deciding what it need not decide there.  One
                    // call site each for ms_fill_simple and ms_fill_from_batch (a second inlined copy costs spills).
                    const bool fast = fast_left != 0u;
                    for (;;) {
                        uint32_t run = 0u;
                        if (fast && samples_clean) run = minu(fast_left, (uint32_t)__builtin_ctz(~(simple_mask >> batch_pos)));
                        if (run != 0u) {
                            uint32_t slot = batch_pos;
                            uint32_t begin = (uint32_t)__builtin_amdgcn_readlane((int)slots.pack, (int)slot) & SLOT_IX_MASK;
                            uint32_t rec_next = rec_pre;
                            if (rec_pre_begin != begin) rec_next = bt.item[minu(begin + lane, MS_ITEM_CAP - 1u)];
                            const uint32_t run_end = batch_pos + run;
                            prof.count(FP_N_FILLS, run);
                            prof.count(FP_N_SIMPLE, run);
                            do {
                                const uint32_t end = ((uint32_t)__builtin_amdgcn_readlane((int)slots.pack, (int)slot) >> SLOT_END_SHIFT) & SLOT_IX_MASK;
                                const int32_t backdrop = __builtin_amdgcn_readlane((int)slots.backdrop, (int)slot);
                                uint32_t rec = rec_next;
                                rec_next = bt.item[minu(end + lane, MS_ITEM_CAP - 1u)];  // the next fill's records start where this one's end
                                if (begin + lane >= end) rec = 0u;
                                begin = end;
                                ms_fill_simple<AA>(sh, bt, sh_samples, slot, backdrop, rec, lane, area, prof);
                                if (mode == MODE_COV) store_cov();
                                {
                                    const float4 c = *reinterpret_cast<const float4 *>(&bt.color[slot][0]);
                                    const vec4 fg{c.x, c.y, c.z, c.w};
#pragma unroll
                                    for (int i = 0; i < 4; i++) src_over(rgba[i], fg, area[i]);
                                }
                                prof.mark(FP_BLEND);
                                slot += 1u;
                            } while (slot != run_end);
                            rec_pre = rec_next;
                            rec_pre_begin = begin;
                            batch_pos = run_end;
                            fast_left -= run;
                            if (fast_left == 0u) break;
                            continue;  // (what follows is not a simple fill: one turn of the general form)
                        }
                        prof.count(FP_N_FILLS, 1u);
                        ms_fill_from_batch<AA>(sh, bt, sh_samples, batch_pos, slots, lane, area, samples_clean, rec_pre, rec_pre_begin, prof);
                        batch_pos += 1u;
                        if (!fast) break;
                        if (mode == MODE_COV) store_cov();
                        // (a slice's wave blends as well: its pixels are never stored, and an unconditional update of rgba
                        // is done in place -- behind a branch the compiler blends into fresh registers and copies all
                        // sixteen back at the loop's join, eight v_mov_b64 per fill)
                        {
                            const float4 c = *reinterpret_cast<const float4 *>(&bt.color[batch_pos - 1u][0]);
                            const vec4 fg{c.x, c.y, c.z, c.w};
#pragma unroll
                            for (int i = 0; i < 4; i++) src_over(rgba[i], fg, area[i]);
                        }
                        prof.mark(FP_BLEND);
                        fast_left -= 1u;
                        if (fast_left == 0u) break;
                    }
                    if (fast) {
                        if (mode == MODE_COV && fill_no >= fill_hi) break;
                        cmd_ix = fast_after;
                        continue;
                    }
                } else {
                    prof.count(FP_N_FILLS, 1u);
                    CmdFill fill;
                    fill.size_and_rule = rd(cmd_ix + 1u);
                    fill.seg_data = rd(cmd_ix + 2u);
                    fill.backdrop = (int32_t)rd(cmd_ix + 3u);
                    fill_path_ms<AA>(sh, sh_samples, segments, mask_lut, fill, lane, area);
                    samples_clean = false;
                    prof.mark(FP_FILL_UNBATCHED);
                }
                if (mode == MODE_COV) {
                    store_cov();
                    cmd_ix += 4u;
                    if (fill_no >= fill_hi) break;
                    continue;
                }
                }
            }
            cmd_ix += 4u;
            // FILL is followed by its draw command, CMD_COLOR as a rule: blend right away instead of going round the loop
            if (cmd_ix + 2u <= win_base + 64u && rd(cmd_ix) == CMD_COLOR) {
                const vec4 fg = unpack4x8unorm(rd(cmd_ix + 1u));
#pragma unroll
                for (int i = 0; i < 4; i++) src_over(rgba[i], fg, area[i]);
                cmd_ix += 2u;
                prof.mark(FP_BLEND);
            }
        } else if (AA != 0 && mode == MODE_COV && tag != CMD_JUMP) {
            // a slice's wave wants the FILLs only (and must keep away from the blend stack's memory: it is the compositing
            // pass's); sizes as the arms below and rare_command advance
            if (tag == CMD_COLOR || tag == CMD_IMAGE) cmd_ix += 2u;
            else if (tag == CMD_END_CLIP || tag == CMD_LIN_GRAD || tag == CMD_RAD_GRAD || tag == CMD_SWEEP_GRAD || tag == CMD_BLUR_RECT) cmd_ix += 3u;
            else cmd_ix += 1u;  // SOLID, BEGIN_CLIP, unknown tags
        } else if (tag == CMD_SOLID) {
#pragma unroll
            for (int i = 0; i < 4; i++) area[i] = 1.0f;
            cmd_ix += 1u;
            if (cmd_ix + 2u <= win_base + 64u && rd(cmd_ix) == CMD_COLOR) {  // (as after FILL)
                const vec4 fg = unpack4x8unorm(rd(cmd_ix + 1u));
#pragma unroll
                for (int i = 0; i < 4; i++) src_over(rgba[i], fg, area[i]);
                cmd_ix += 2u;
            }
        } else if (tag == CMD_COLOR) {
            const vec4 fg = unpack4x8unorm(rd(cmd_ix + 1u));
#pragma unroll
            for (int i = 0; i < 4; i++) src_over(rgba[i], fg, area[i]);
            cmd_ix += 2u;
        } else if (tag == CMD_JUMP) {
            cmd_ix = rd(cmd_ix + 1u);
        } else {
            RareState st;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                st.rgba[i] = rgba[i];
                st.area[i] = area[i];
            }
            st.clip_depth = clip_depth;
            st.cmd_ix = cmd_ix;
            if constexpr (BRUSHES)
                rare_command_body<BRUSHES>(st, blend_stack, tag, cfg.ptcl_size, cfg.blend_size, ptcl, info, blend_spill, blend_offset, lane, xy_x, xy_y, ramps,
                                           n_ramps, atlas_texels, atlas_w, atlas_h);
            else
                rare_command<BRUSHES, WAVES>(st, blend_stack, tag, cfg.ptcl_size, cfg.blend_size, ptcl, info, blend_spill, blend_offset, lane, xy_x, xy_y, ramps, n_ramps,
                                      atlas_texels, atlas_w, atlas_h);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                rgba[i] = st.rgba[i];
                area[i] = st.area[i];
            }
            clip_depth = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.clip_depth);
            cmd_ix = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.cmd_ix);
            prof.mark(tag == CMD_END_CLIP ? FP_RARE_END_CLIP
                      : tag == CMD_LIN_GRAD || tag == CMD_RAD_GRAD || tag == CMD_SWEEP_GRAD ? FP_RARE_GRAD
                      : tag == CMD_IMAGE || tag == CMD_BLUR_RECT ? FP_RARE_IMAGE : FP_RARE);
            prof.count(FP_N_RARE, 1u);
        }
    }
    if (AA != 0 && mode == MODE_COV) {
    // The slice is done: take a ticket.  The in-launch hand-off of cdna_hip_programming.md section 6 guideline 16 in its
    // write-through form: the coverage went out with agent-scope (sc1) stores, this wave drains them, then a relaxed
    // agent-scope ticket; the wave that draws the last ticket reads every slice's bytes with agent-scope loads.  No
    // release / acquire fences: a fence per slice writes back and invalidates whole caches under the other frames'
    // kernels (measured: -15 % frames/s with four frames in flight).  Correct wherever the slices ran (other CUs, other
    // XCDs); the counter was zeroed by coarse.
    // What this relies on -- sc1 stores written through to where every XCD's agent-scope load finds them, completed when vmcnt
    // says so -- is gfx950's behaviour, not the HSA memory model's promise: the device pass refuses to build for anything else
    // (VERDICT r4 item 8), and tests/test_gpu_parity.py::test_fine_slice_handoff_stress_across_xcds holds it to the unsliced
    // image with every tile's slices on different XCDs and four frames in flight.
#ifndef VELLO_SIMT_EMU
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_fine's slice hand-off (write-through stores + relaxed ticket, no fences) is validated for gfx950 only"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    uint32_t ticket = 0u;
    if (lane == 0u) ticket = __hip_atomic_fetch_add(&slice_counters[first_item], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
    if (ticket + 1u != slice_n) {
        tl.log(blend_spill, cfg.blend_size, lane, 1u, fill_no, tile_ix);
        return;
    }
    // The tile's last slice has arrived: composite (out of line: the second interpreter must not cost the first its registers)
    {
        BlendPassOut bo;
        blend_pass<AA, BRUSHES, WAVES>(bo, tile_ix, cfg.base_color, cfg.ptcl_size, cfg.blend_size, ptcl, info, blend_spill, cov + cov_base, sh_samples, fill_room, lane, xy_x, xy_y,
                                ramps, n_ramps, atlas_texels, atlas_w, atlas_h);
#pragma unroll
        for (int i = 0; i < 4; i++) rgba[i] = bo.rgba[i];
    }
    }
    prof.mark(FP_INTERP);
    prof.store(blend_spill, cfg.blend_size, n_tiles, tile_ix, lane);
    tl.log(blend_spill, cfg.blend_size, lane, AA != 0 && mode == MODE_COV ? 2u : 0u, fill_no, tile_ix);
    // fine.wgsl:1386-1397: un-premultiplied RGBA8
    const uint32_t px0 = tile_x * TILE_WIDTH + lx * PIXELS_PER_THREAD;
    const uint32_t py = tile_y * TILE_HEIGHT + ly;
    if (py < cfg.target_height && px0 < cfg.target_width) {
        uint32_t packed[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            vec4 fg = rgba[i];
            float a_inv = 1.0f / maxf(fg.w, 1e-6f);
            packed[i] = pack4x8unorm(vec4{fg.x * a_inv, fg.y * a_inv, fg.z * a_inv, fg.w});
        }
        uint8_t *row = output + (size_t)py * out_stride + (size_t)px0 * 4u;
        if (px0 + 4u <= cfg.target_width && ((reinterpret_cast<uintptr_t>(row) & 15u) == 0u)) {
            *reinterpret_cast<uint4 *>(row) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++)
                if (px0 + i < cfg.target_width) reinterpret_cast<uint32_t *>(row)[i] = packed[i];
        }
    }
}


template <int AA>
static void launch_fine_aa(const Frame &f, hipStream_t s, const uint32_t *mask_lut) {
    // MSAA: slice_cap blocks for the slices of long tiles come first (blocks beyond the frame's count leave at once)
    const uint32_t slice_blocks = AA != 0 && f.slice_min_fills != 0u ? f.slice_cap : 0u;
    dim3 grid(f.cfg.width_in_tiles * f.cfg.height_in_tiles + slice_blocks);
    uint32_t stride = (uint32_t)f.out_stride;
    constexpr int WAVES_IN_FLIGHT = AA != 0 ? VK_FINE_WAVES_IN_FLIGHT : 4;  // (area AA: one form)
    if (f.brushes)
        hipLaunchKernelGGL((k_fine<AA, true>), grid, dim3(64), 0, s, f.cfg, f.segments, f.ptcl, f.info_bin_data, f.blend_spill, f.output,
                           stride, f.ramps, f.n_ramps, mask_lut, f.atlas, f.atlas_w, f.atlas_h, f.control->work_count, f.tile_order, f.slice_items,
                           f.slice_counters, f.cov, slice_blocks, f.slice_fills);
    else if (WAVES_IN_FLIGHT != 4 && !f.flatten_side_by_side)  // (frames in flight)
            hipLaunchKernelGGL((k_fine<AA, false, WAVES_IN_FLIGHT>), grid, dim3(64), 0, s, f.cfg, f.segments, f.ptcl, f.info_bin_data, f.blend_spill, f.output,
                               stride, f.ramps, f.n_ramps, mask_lut, f.atlas, f.atlas_w, f.atlas_h, f.control->work_count, f.tile_order, f.slice_items,
                               f.slice_counters, f.cov, slice_blocks, f.slice_fills);
    else
        hipLaunchKernelGGL((k_fine<AA, false>), grid, dim3(64), 0, s, f.cfg, f.segments, f.ptcl, f.info_bin_data, f.blend_spill, f.output,
                           stride, f.ramps, f.n_ramps, mask_lut, f.atlas, f.atlas_w, f.atlas_h, f.control->work_count, f.tile_order, f.slice_items,
                           f.slice_counters, f.cov, slice_blocks, f.slice_fills);
}

void launch_fine(const Frame &f, hipStream_t s) {
    if (f.cfg.width_in_tiles * f.cfg.height_in_tiles == 0) return;
    if (f.aa == 0) launch_fine_aa<0>(f, s, f.mask_lut8);
    else if (f.aa == 1) launch_fine_aa<1>(f, s, f.mask_lut8);
    else launch_fine_aa<2>(f, s, f.mask_lut16);
}

}  // namespace vk
