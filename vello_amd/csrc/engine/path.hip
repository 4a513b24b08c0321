// path_count (+setup), backdrop_dyn, path_tiling (+setup).
// Reference: vello_shaders/shader/path_count_setup.wgsl:17-27, path_count.wgsl:51-202,
// backdrop_dyn.wgsl:28-86, path_tiling_setup.wgsl:20-32, path_tiling.wgsl:39-173
// (vello/src/render.rs:443-502); CPU twins cpu/{path_count,backdrop,path_tiling}.rs.
//
// HIP has no indirect dispatch: both line-driven kernels launch a fixed grid and stride over the
// count the previous stage left in `bump` (the *_setup dispatches disappear).
#include "engine.h"

namespace vk {

namespace {

// Everything path_count derives from one line before it walks tiles (path_count.wgsl:60-164).
struct LineWalk {
    bool valid;
    bool is_down, is_positive_slope;
    float a, b, x0, y0, x_sign, s0x, s0y;
    uint32_t imin, imax;
    int32_t ymin, ymax;
    int32_t bbox0, bbox1, bbox2, bbox3, stride;
    uint32_t tiles_base;
};

// A LineSoup record (24 B, 8-byte aligned) and a Path record (32 B) as whole-record vector loads: left to itself the
// compiler fetches the fields one dword at a time, each under the branch that first needs it (8 + 5 load
// instructions per line and pass).
struct __attribute__((aligned(8))) Words2 { uint32_t a, b; };
struct __attribute__((aligned(16))) Words4 { uint32_t a, b, c, d; };
__device__ __forceinline__ LineSoup load_line(const LineSoup *__restrict__ lines, uint32_t ix) {
    const Words2 *p = reinterpret_cast<const Words2 *>(lines + ix);
    const Words2 w0 = p[0], w1 = p[1], w2 = p[2];
    LineSoup l;
    l.path_ix = w0.a; l.pad = w0.b;
    l.p0x = __uint_as_float(w1.a); l.p0y = __uint_as_float(w1.b);
    l.p1x = __uint_as_float(w2.a); l.p1y = __uint_as_float(w2.b);
    return l;
}
__device__ __forceinline__ Path load_path(const Path *__restrict__ paths, uint32_t ix) {
    const Words4 *p = reinterpret_cast<const Words4 *>(paths + ix);
    const Words4 w0 = p[0];
    Path r;
    r.bbox[0] = w0.a; r.bbox[1] = w0.b; r.bbox[2] = w0.c; r.bbox[3] = w0.d;
    r.tiles = paths[ix].tiles;
    r.pad[0] = r.pad[1] = r.pad[2] = 0u;
    return r;
}

// `get_path`: the line's Path record, asked for only once the line is known to cross anything (pass 1 loads it and sets it
// aside, pass 2 takes it from there: k_path_count below)
// Written without early exits: a line that crosses nothing carries `valid == false` through the same arithmetic (its
// values are never used), so that the compiler keeps one copy of the twenty results instead of moving them between the
// copies of four exit paths (100 of the 212 VALU instructions per line were v_mov).  `get_path` is asked for every line.
template <class GetPath>
__device__ __forceinline__ LineWalk setup_line_walk(const LineSoup &line, GetPath get_path, uint32_t n_paths) {
    LineWalk w;
    // A tag stream with more PATH markers than the layout counts (only a hand-made stream: resolve appends its extra
    // markers behind the last segment, resolve.rs:127-129) yields lines whose path has no Path record.  WebGPU reads
    // zeros there (stride 0 -> no crossings, path_count.wgsl:112); HIP would read past paths[]: get_path clamps.
    bool valid = line.path_ix < n_paths;
    const float TILE_SCALE = 0.0625f;
    bool is_down = line.p1y >= line.p0y;
    vec2 xy0 = is_down ? v2(line.p0x, line.p0y) : v2(line.p1x, line.p1y);
    vec2 xy1 = is_down ? v2(line.p1x, line.p1y) : v2(line.p0x, line.p0y);
    vec2 s0 = xy0 * TILE_SCALE;
    vec2 s1 = xy1 * TILE_SCALE;
    uint32_t count_x = span(s0.x, s1.x) - 1u;
    uint32_t count = count_x + span(s0.y, s1.y);
    float dx = fabsf(s1.x - s0.x);
    float dy = s1.y - s0.y;
    valid = valid && !(dx + dy == 0.0f) && !(dy == 0.0f && floorf(s0.y) == s0.y);
    float idxdy = 1.0f / (dx + dy);
    float a = dx * idxdy;
    bool is_positive_slope = s1.x >= s0.x;
    float x_sign = is_positive_slope ? 1.0f : -1.0f;
    float xt0 = floorf(s0.x * x_sign);
    float c = s0.x * x_sign - xt0;
    float y0 = floorf(s0.y);
    float ytop = (s0.y == s1.y) ? ceilf(s0.y) : y0 + 1.0f;
    float b = minf((dy * c + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
    float robust_err = floorf(a * ((float)count - 1.0f) + b) - (float)count_x;
    if (robust_err != 0.0f) a -= ROBUST_EPSILON * signf(robust_err);
    float x0 = xt0 * x_sign + (is_positive_slope ? 0.0f : -1.0f);

    Path path = get_path();
    int32_t bbox0 = (int32_t)path.bbox[0], bbox1 = (int32_t)path.bbox[1], bbox2 = (int32_t)path.bbox[2], bbox3 = (int32_t)path.bbox[3];
    float xmin = minf(s0.x, s1.x);
    int32_t stride = bbox2 - bbox0;
    valid = valid && !(s0.y >= (float)bbox3 || s1.y <= (float)bbox1 || xmin >= (float)bbox2 || stride == 0);
    uint32_t imin = 0u;
    if (s0.y < (float)bbox1) {
        float iminf = roundf_te(((float)bbox1 - y0 + b - a) / (1.0f - a)) - 1.0f;
        if (y0 + iminf - floorf(a * iminf + b) < (float)bbox1) iminf += 1.0f;
        imin = f2u(iminf);
    }
    uint32_t imax = count;
    if (s1.y > (float)bbox3) {
        float imaxf = roundf_te(((float)bbox3 - y0 + b - a) / (1.0f - a)) - 1.0f;
        if (y0 + imaxf - floorf(a * imaxf + b) < (float)bbox3) imaxf += 1.0f;
        imax = f2u(imaxf);
    }
    int32_t ymin = 0, ymax = 0;
    if (maxf(s0.x, s1.x) <= (float)bbox0) {
        ymin = f2i(ceilf(s0.y));
        ymax = f2i(ceilf(s1.y));
        imax = imin;
    } else {
        float fudge = is_positive_slope ? 0.0f : 1.0f;
        if (xmin < (float)bbox0) {
            float f = roundf_te((x_sign * ((float)bbox0 - x0) - b + fudge) / a);
            if ((x0 + x_sign * floorf(a * f + b) < (float)bbox0) == is_positive_slope) f += 1.0f;
            int32_t ynext = f2i(y0 + f - floorf(a * f + b) + 1.0f);
            if (is_positive_slope) {
                if (f2u(f) > imin) {
                    ymin = f2i(y0 + (y0 == s0.y ? 0.0f : 1.0f));
                    ymax = ynext;
                    imin = f2u(f);
                }
            } else {
                if (f2u(f) < imax) {
                    ymin = ynext;
                    ymax = f2i(ceilf(s1.y));
                    imax = f2u(f);
                }
            }
        }
        if (maxf(s0.x, s1.x) > (float)bbox2) {
            float f = roundf_te((x_sign * ((float)bbox2 - x0) - b + fudge) / a);
            if ((x0 + x_sign * floorf(a * f + b) < (float)bbox2) == is_positive_slope) f += 1.0f;
            if (is_positive_slope) imax = minu(imax, f2u(f));
            else imin = maxu(imin, f2u(f));
        }
    }
    imax = maxu(imin, imax);
    w.valid = valid;
    w.is_down = is_down;
    w.is_positive_slope = is_positive_slope;
    w.a = a; w.b = b; w.x0 = x0; w.y0 = y0; w.x_sign = x_sign; w.s0x = s0.x; w.s0y = s0.y;
    w.imin = valid ? imin : 0u; w.imax = valid ? imax : 0u;
    w.ymin = valid ? maxi(ymin, bbox1) : 0;
    w.ymax = valid ? mini(ymax, bbox3) : 0;
    w.bbox0 = bbox0; w.bbox1 = bbox1; w.bbox2 = bbox2; w.bbox3 = bbox3; w.stride = stride;
    w.tiles_base = path.tiles;
    return w;
}

}  // namespace

// ---- k_path_count: path_count.wgsl:51-202, the tile atomics of a whole workgroup chunk added up in LDS first ------------------
//
// A workgroup takes chunks of 256 x LPT lines; the reference is a thread per line, an atomicAdd(bump.seg_counts) per line and
// one on the tile per crossing.  Here: one bump atomic per chunk, and the tile atomics of the chunk added up in LDS first.
//
// What a scattered atomic costs is one request per (instruction, distinct cache line), 2.7e10 per second chip-wide, whether
// the lines come from one XCD or from all (scripts/calib/atomic_rate.hip, atomic_scope.hip).  A returning add per run of
// crossings on one tile (rounds 1-3) was 2.0-2.3 M requests on the road map (scripts/pc_requests.py) and ran at that rate.
// But the 1 024 lines of a chunk are consecutive lines of a few paths and cross the same tiles over and over: per chunk 1 430
// crossings fall into 475 tiles on 207 cache lines of the tile pool.  So the workgroup counts in LDS -- a small hash table keyed
// by the cache line (16 tiles), a word per tile: its crossings in the low half; in the high half the top-edge backdrop bumps of
// the tile to its right, so that a crossing is one add -- and then asks memory ONCE per touched tile, 16 lanes on the 16 tiles
// of a line: 0.7-0.9 M requests.  The returned old value of a tile becomes the cursor its crossings draw their slot index from
// (a returning LDS add).  Which crossing of a tile gets which slot is as arbitrary as in the reference (path_count.wgsl:189 is
// an atomicAdd in dispatch order).  A tile whose cache line finds no place in the table (PC_PROBES places taken by other
// lines) goes to memory directly, as every crossing used to.
//
// The walks stay in the registers of the threads that set them up (9 words a line).  The counting pass is a thread per line;
// what a crossing learns there -- the address of its cursor, or the slot index itself when it went to memory directly --
// waits in a per-wave stash (PC_STASH crossings; the ones beyond it take the direct route afterwards), so that the
// SegmentCount records are written by a pass with a crossing per lane that repeats none of the arithmetic.
#ifndef VK_PC_TABLE_LOG2
#define VK_PC_TABLE_LOG2 9
#endif
#ifndef VK_PC_GRID
// the launch: as many workgroups as the chip holds at once (three per CU), striding over the chunks -- a grid sized for the pool's
// capacity is thousands of workgroups that find no chunk, and each is dispatched with its 48 KB of LDS before it can say so
// (r1mix one frame at a time +2.6 %, d2 four in flight +0.9 %, profiles/r04_ab_s18_constants.txt)
#define VK_PC_GRID 768u
#endif
#ifndef VK_PC_STASH
#define VK_PC_STASH 768
#endif
// Round 6: the kernel's LDS footprint is a parameter.  With frames in flight k_path_count shares the CUs with the other frames'
// kernels, and 3 x 48.7 KB left them 14 KB of a CU's LDS for as long as its workgroups waited for their flushes (266 us under
// overlap for 103 alone, profiles/r05_pipeline_timeline.txt).  PcInFlight -- chunks of 512 lines, a table of 256 cache lines, a
// stash of 384 crossings per wave: 24.6 KB, 88 VGPRs -- is 12 % slower alone (118 against 105 us on d2) and +3.6 % frames/s with
// four frames in flight (a process per build, alternating: profiles/r06_ab_path_count_footprint.txt); one frame at a time keeps
// the large form.
#ifndef VK_PC_TABLE_LOG2_IN_FLIGHT
#define VK_PC_TABLE_LOG2_IN_FLIGHT 8
#endif
#ifndef VK_PC_STASH_IN_FLIGHT
#define VK_PC_STASH_IN_FLIGHT 384
#endif
#ifndef VK_PC_LPT_IN_FLIGHT
#define VK_PC_LPT_IN_FLIGHT 2
#endif
#ifndef VK_PC_GRID_IN_FLIGHT
#define VK_PC_GRID_IN_FLIGHT 768u
#endif
#ifndef VK_PC_FLUSH_GROUP
#define VK_PC_FLUSH_GROUP 4
#endif
constexpr uint32_t PC_FLUSH_GROUP = VK_PC_FLUSH_GROUP;  // turns of the flush loop whose returning adds are in flight together
constexpr uint32_t PC_PROBES = 3u;
constexpr uint32_t PC_EMPTY = 0xffffffffu, PC_NONE = 0xffffffffu;
constexpr uint32_t PC_DONE = 0x80000000u;   // stash word: bits 0-15 are the slot index already (else: the cursor's word index)
template <uint32_t TABLE_LOG2_, uint32_t STASH_>
struct PcParams {
    static constexpr uint32_t TABLE_LOG2 = TABLE_LOG2_, TABLE = 1u << TABLE_LOG2_;
    static constexpr uint32_t CNT_WORDS = TABLE * 16u;
    static constexpr uint32_t STASH = STASH_;  // crossings per wave and chunk that wait in LDS for their cursors
    // The packed fields below hold only for these sizes (the constants are -D-overridable for sweeps and for
    // scripts/emu_variant_check.sh; a value beyond them would corrupt slots and backdrops silently instead of failing the build):
    static_assert(CNT_WORDS + 64u <= 65536u, "a stash word keeps the cursor's word index (spare words included) in bits 0-15");
    static_assert(TABLE <= 65536u, "PcShared::occupied is uint16_t");
    static_assert(4u * STASH < 32768u, "a cnt word: crossings in bits 0-15, the SIGNED sum of backdrop bumps above -- both bounded by the chunk's stashed crossings (4 waves x STASH)");
};
using PcAlone = PcParams<VK_PC_TABLE_LOG2, VK_PC_STASH>;
using PcInFlight = PcParams<VK_PC_TABLE_LOG2_IN_FLIGHT, VK_PC_STASH_IN_FLIGHT>;

// `c` for some lane of the wave, as a scalar branch condition: the seldom-taken arms below are a handful of instructions, which
// the compiler would run with an empty exec mask rather than branch over.  (The emulator runs lanes as fibers: the lane's own c.)
#ifdef VELLO_SIMT_EMU
#define PC_WAVE_ANY(c) (c)
#else
#define PC_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#endif

template <class P>
struct PcShared {
    // (the 64 words behind keys[] and cnt[]: a place of its own for every lane whose LDS operation is not wanted -- the walk
    // loop selects ADDRESSES instead of branching around the operations: a branch costs five scalar instructions, a select one)
    uint32_t keys[P::TABLE + 64u];     // tile index >> 4 of the entry, PC_EMPTY; the spare words hold PC_NOBODY
    // counting, a word per tile: its crossings | << 16 the sum of the backdrop bumps of the tile to its RIGHT (where a crossing
    // with a top edge puts its bump, path_count.wgsl:181-186: one add does both); after the flush: the tile's slot cursor
    uint32_t cnt[P::CNT_WORDS + 64u];
    uint16_t occupied[P::TABLE];
    uint32_t n_occ;
    uint32_t stash[4][P::STASH]; // cursor word index or PC_DONE | slot index; | lane of the line << 16
    uint32_t wave_total[4];
    uint32_t base;
};
constexpr uint32_t PC_NOBODY = 0xfffffffeu;
static_assert(PC_PROBES >= 1u, "an entry is looked for in at least one place");

struct PcWalk {
    float a, b, x0, y0;
    uint32_t ioff, count;        // crossing index of the line's item q (counted over the wave's 64 lines): ioff + q
    uint32_t base;               // tile index of (x, y) = base + y * stride + x
    uint32_t bbox02;             // bbox0 | bbox2 << 16 (tile coordinates of a target of at most 65 535 tiles, engine.hip)
    uint32_t flags;              // 1: is_down, 2: negative slope, 4: y0 == s0.y
};

template <class P>
__device__ __forceinline__ uint32_t pc_hash(uint32_t line) { return (line * 0x9E3779B1u) >> (32u - P::TABLE_LOG2); }
// the places a cache line of tiles may take when its first one (pc_hash) is somebody else's; PC_NONE if they all are (the table
// only grows within a chunk: whoever asks for the same line later gets the same answer)
template <class P>
__device__ __forceinline__ uint32_t pc_slot_more(PcShared<P> &sh, uint32_t line) {
    uint32_t h = pc_hash<P>(line);
    const uint32_t step = (line >> 2) | 1u;
#pragma unroll 1
    for (uint32_t p = 1; p < PC_PROBES; p++) {
        h = (h + step) & (P::TABLE - 1u);
        const uint32_t o = atomicCAS(&sh.keys[h], PC_EMPTY, line);
        if (o == PC_EMPTY || o == line) return h;
    }
    return PC_NONE;
}

// The counting pass (path_count.wgsl:172-199): the first n_stash crossings of every line of the wave into the table and the stash;
// the line's crossing s is item `item0 + s` of the wave.
//
// A thread per line, like the reference -- and then the pass lasts as long as the wave's LONGEST line: on a road map lines cross
// 1.4 tiles, but a small scene's few workgroups wait for the one whose chunk holds the edges of a background rectangle (the tiger:
// chunks take 5.8 us, one takes 30 -- a 20 us counting pass over 64-tile lines -- and that one is the launch;
// scripts/pc_timeline.py tiger).  Crossings are independent of each other (z of crossing i is floor(a i + b), the z before it
// floor(a (i - 1) + b): the same two roundings the loop carries along), so when a wave holds FEW long lines their crossings beyond
// the first PC_COOP_FROM are spread over the lanes, 64 a round, line by line: the walk's nine words are read from the owner's lane
// (v_readlane), the stash word names the owner as before.  "Few": the rounds that takes (twice over, for the reads) against the steps
// the longest line would still walk alone -- a wave of mmark's lines, every one of them 80 tiles long, keeps a thread per line.
#ifndef VK_PC_COOP_FROM
#define VK_PC_COOP_FROM 8u
#endif
constexpr uint32_t PC_COOP_FROM = VK_PC_COOP_FROM;
template <class P>
__device__ __forceinline__ void pc_count_lines(PcShared<P> &sh, const PcWalk &w, uint32_t imin, uint32_t item0, uint32_t n_stash, uint32_t lane, uint32_t wave,
                                               const Config &cfg, Tile *tile) {
    // crossing i of the line with this walk, owned by lane `owner`, item `item` of the wave
    auto crossing = [&](float a, float b, float x0, float y0, uint32_t base, uint32_t bbox02, uint32_t flags, uint32_t i, float last_z, float z,
                        uint32_t item, uint32_t owner) {
        (void)a; (void)b;
        const int32_t bbox0 = (int32_t)(bbox02 & 0xffffu), bbox2 = (int32_t)(bbox02 >> 16);
        const uint32_t stride = (uint32_t)(bbox2 - bbox0);
        const float x_sign = (flags & 2u) ? -1.0f : 1.0f;
        const int32_t delta = (flags & 1u) ? -1 : 1;
        const int32_t y = f2i(y0 + (float)i - z);
        const int32_t x = f2i(x0 + x_sign * z);
        const uint32_t row = base + (uint32_t)y * stride;
        const bool top_edge = i == 0u ? (flags & 4u) != 0u : last_z == z;  // (not folded into last_z: z is a NaN when a is not finite)
        const uint32_t key = row + (uint32_t)x;
        const int32_t x1 = (int32_t)((uint32_t)x + 1u);  // (x saturates at the ends of i32 for coordinates like 3e38: WGSL's i32 wraps)
        const uint32_t bkey = row + (uint32_t)maxi(x1, bbox0);
        const bool counted = key < cfg.tiles_size;
        const bool bump = top_edge && x1 < bbox2 && bkey < cfg.tiles_size;
        // the usual crossing: in the pool, its cache line in the first place the table gives it, its bump (if any) on the tile to
        // its right -- a compare-and-swap, an add and the stash, no branch
        const uint32_t line = key >> 4, h = pc_hash<P>(line);
        const uint32_t o = atomicCAS(&sh.keys[counted ? h : P::TABLE + lane], PC_EMPTY, line);
        const bool hit = o == PC_EMPTY || o == line;  // (a lane outside the pool reads PC_NOBODY)
        const bool bump_right = bump && bkey == key + 1u;
        const uint32_t at = h * 16u + (key & 15u);
        const uint32_t one = 1u + (bump_right ? (uint32_t)delta << 16 : 0u);
        atomicAdd(&sh.cnt[hit ? at : P::CNT_WORDS + lane], one);
        uint32_t word = hit ? at : PC_DONE;  // (a crossing outside the pool gets slot 0, as a robust access would give it)
        // everything else, seldom: another place in the table, or none and straight to memory; a bump that is not on the right
        const bool rest = (counted && !hit) || (bump && !(hit && bump_right));
        if (PC_WAVE_ANY(rest)) {
            if (rest) {
                bool bumped = hit && bump_right;
                if (counted && !hit) {
                    const uint32_t slot = pc_slot_more<P>(sh, line);
                    if (slot != PC_NONE) {
                        word = slot * 16u + (key & 15u);
                        atomicAdd(&sh.cnt[word], one);
                        bumped = bump_right;
                    } else {
                        word = PC_DONE | (atomicAdd(&tile[key].segment_count_or_ix, 1u) & 0xffffu);
                    }
                }
                if (bump && !bumped) atomicAdd(&tile[bkey].backdrop, delta);
            }
        }
        sh.stash[wave][item] = word | (owner << 16);
    };
    // few long lines in the wave?  (everything here is wave-uniform)
    uint32_t n_own = n_stash;
    const unsigned long long m_long = __ballot(n_stash > PC_COOP_FROM);
    bool coop = false;
    if (m_long != 0ull) {
        const uint32_t longest = wave_read(wave_incl_scan_max_u32(n_stash, (int)lane), 63u);
        const uint32_t rounds = n_stash > PC_COOP_FROM ? (n_stash - PC_COOP_FROM + 63u) / 64u : 0u;
        const uint32_t all_rounds = wave_read(wave_incl_scan_u32(rounds, (int)lane), 63u);
        coop = all_rounds * 2u + 2u < longest - PC_COOP_FROM;
        if (coop) n_own = minu(n_stash, PC_COOP_FROM);
    }
    // a thread per line.  last_z: the z of the crossing before (path_count.wgsl:172-186 carries it through its loop)
    float last_z = floorf(w.a * ((float)imin - 1.0f) + w.b);
    for (uint32_t s = 0u; s < n_own; s++) {
        const uint32_t i = imin + s;
        const float z = floorf(w.a * (float)i + w.b);
        crossing(w.a, w.b, w.x0, w.y0, w.base, w.bbox02, w.flags, i, last_z, z, item0 + s, lane);
        last_z = z;
    }
    // the long lines' other crossings, a lane per crossing
    if (coop) {
        for (unsigned long long m = m_long; m != 0ull; m &= m - 1ull) {
            const uint32_t owner = (uint32_t)__ffsll((long long)m) - 1u;
            const float a = __uint_as_float(wave_read(__float_as_uint(w.a), owner)), b = __uint_as_float(wave_read(__float_as_uint(w.b), owner));
            const float x0 = __uint_as_float(wave_read(__float_as_uint(w.x0), owner)), y0 = __uint_as_float(wave_read(__float_as_uint(w.y0), owner));
            const uint32_t base = wave_read(w.base, owner), bbox02 = wave_read(w.bbox02, owner), flags = wave_read(w.flags, owner);
            const uint32_t imin_o = wave_read(imin, owner), item_o = wave_read(item0, owner), n_o = wave_read(n_stash, owner);
            for (uint32_t s = PC_COOP_FROM + lane; s < n_o; s += 64u) {
                const uint32_t i = imin_o + s;
                const float z = floorf(a * (float)i + b);
                const float z_before = floorf(a * (float)(i - 1u) + b);
                crossing(a, b, x0, y0, base, bbox02, flags, i, z_before, z, item_o + s, owner);
            }
        }
    }
}

// The crossings of one line from its item `n_stash` on -- the ones without a place in the stash -- a thread per line, straight to
// memory and to the SegmentCount pool (the wave's records start at `seg_wave`), as the reference does every crossing.
__device__ __forceinline__ void pc_walk_line_direct(const PcWalk &w, uint32_t imin, uint32_t item0, uint32_t n_stash, const Config &cfg, Tile *tile,
                                                    uint32_t seg_wave, uint32_t line_ix, SegmentCount *__restrict__ seg_counts) {
    const int32_t bbox0 = (int32_t)(w.bbox02 & 0xffffu), bbox2 = (int32_t)(w.bbox02 >> 16);
    const uint32_t stride = (uint32_t)(bbox2 - bbox0);
    const float x_sign = (w.flags & 2u) ? -1.0f : 1.0f;
    const int32_t delta = (w.flags & 1u) ? -1 : 1;
    const uint32_t s_begin = n_stash, s_end = w.count;
    const uint32_t i_begin = imin + s_begin;
    float last_z = floorf(w.a * (s_begin == 0u ? (float)imin - 1.0f : (float)(i_begin - 1u)) + w.b);
    const bool top_edge_0 = (w.flags & 4u) != 0u;
    for (uint32_t s = s_begin; s < s_end; s++) {
        const uint32_t i = imin + s;
        const float z = floorf(w.a * (float)i + w.b);
        const int32_t y = f2i(w.y0 + (float)i - z);
        const int32_t x = f2i(w.x0 + x_sign * z);
        const uint32_t row = w.base + (uint32_t)y * stride;
        const bool top_edge = i == 0u ? top_edge_0 : last_z == z;  // (not folded into last_z: z is a NaN when a is not finite)
        last_z = z;
        const uint32_t key = row + (uint32_t)x;
        const int32_t x1 = (int32_t)((uint32_t)x + 1u);  // (x saturates at the ends of i32 for coordinates like 3e38: WGSL's i32 wraps)
        const uint32_t bkey = row + (uint32_t)maxi(x1, bbox0);
        const bool counted = key < cfg.tiles_size;
        const bool bump = top_edge && x1 < bbox2 && bkey < cfg.tiles_size;
        uint32_t seg_within_slice = 0u;
        if (counted) seg_within_slice = atomicAdd(&tile[key].segment_count_or_ix, 1u);
        if (bump) atomicAdd(&tile[bkey].backdrop, delta);
        const uint32_t seg_ix = seg_wave + item0 + s;
        if (seg_ix < cfg.seg_counts_size) {
            SegmentCount sc;
            sc.line_ix = line_ix;
            sc.counts = (seg_within_slice << 16) | i;
            seg_counts[seg_ix] = sc;
        }
    }
}

template <uint32_t LPT, class P>
__global__ void __launch_bounds__(256) k_path_count(Config cfg, Bump *bump, const LineSoup *__restrict__ lines,
                                                        const Path *__restrict__ paths, Tile *tile, SegmentCount *__restrict__ seg_counts) {
    __shared__ PcShared<P> sh;
    constexpr uint32_t PC_TABLE = P::TABLE, PC_CNT_WORDS = P::CNT_WORDS, PC_STASH = P::STASH;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (bump->failed != 0u) return;  // path_count_setup.wgsl:18-19
    const uint32_t n_lines = minu(bump->lines, cfg.lines_size);
    const uint32_t n_paths = cfg.layout.n_paths;
    constexpr uint32_t CHUNK = 256u * LPT;
    if (blockIdx.x * CHUNK >= n_lines) return;  // (the grid is sized for the pool: most workgroups of a small scene)
    for (uint32_t k = tid; k < PC_TABLE + 64u; k += 256u) sh.keys[k] = k < PC_TABLE ? PC_EMPTY : PC_NOBODY;
    for (uint32_t k = tid; k < PC_CNT_WORDS + 64u; k += 256u) sh.cnt[k] = 0u;
    if (tid == 0u) sh.n_occ = 0u;
    __syncthreads();
    for (uint32_t chunk = blockIdx.x * CHUNK; chunk < n_lines; chunk += gridDim.x * CHUNK) {
#ifdef VELLO_PC_TIMELINE
        // measurement build (scripts/pc_timeline.py): wall-clock stamps (100 MHz) of the chunk's phases in the tail of the pool
        const uint32_t tl0 = (uint32_t)wall_clock64();
#endif
        // ---- the walks; the wave's and the chunk's slices of the SegmentCount pool ----
        LineSoup ln[LPT];
        Path pa[LPT];
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) ln[j] = load_line(lines, minu(chunk + j * 256u + tid, n_lines - 1u));
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) pa[j] = load_path(paths, ln[j].path_ix < n_paths ? ln[j].path_ix : 0u);  // (the pool holds >= 256 records)
        PcWalk w[LPT];
        uint32_t p[LPT], T[LPT], wave_total = 0u;
        // Places in the stash: item `wave_total + p` onwards while they last.  The sums are u32 like the reference's bump counter
        // and wrap with lines of billions of crossings (coordinates like 3e38): from the first line of the wave with more
        // crossings than the stash holds, nobody's item number is trusted for a place (V[j]: the stashed items of line group j).
        uint32_t n_stash[LPT], V[LPT];
        bool trusted = true;  // (wave-uniform)
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) {
            LineWalk lw = setup_line_walk(ln[j], [&]() { return pa[j]; }, chunk + j * 256u + tid < n_lines ? n_paths : 0u);
            w[j].a = lw.a; w[j].b = lw.b; w[j].x0 = lw.x0; w[j].y0 = lw.y0;
            w[j].count = lw.imax - lw.imin;  // (0 for a line that crosses nothing)
            w[j].base = lw.tiles_base - (uint32_t)lw.bbox1 * (uint32_t)lw.stride - (uint32_t)lw.bbox0;
            w[j].bbox02 = ((uint32_t)lw.bbox0 & 0xffffu) | ((uint32_t)lw.bbox2 << 16);
            w[j].flags = (lw.is_down ? 1u : 0u) | (lw.is_positive_slope ? 0u : 2u) | (lw.y0 == lw.s0y ? 4u : 0u);
            const uint32_t incl = wave_incl_scan_u32(w[j].count, (int)lane);
            p[j] = incl - w[j].count;
            T[j] = wave_read(incl, 63u);
            w[j].ioff = lw.imin - p[j];
            {
                const unsigned long long big = __ballot(w[j].count > PC_STASH);
                const uint32_t first_big = big ? (uint32_t)__ffsll((long long)big) - 1u : 64u;
                const uint32_t item0 = wave_total + p[j];
                const bool mine = trusted && lane <= first_big && item0 < PC_STASH;
                n_stash[j] = mine ? minu(w[j].count, PC_STASH - item0) : 0u;
                const uint32_t upto = first_big < 64u ? wave_read(p[j] + n_stash[j], first_big) : T[j];  // (items of the group before the doubt)
                V[j] = trusted && wave_total < PC_STASH ? minu(upto, PC_STASH - wave_total) : 0u;
                trusted = trusted && first_big == 64u;
            }
            wave_total += T[j];
            // (a line left of its path's rectangle: a bump in column 0 of every row it spans -- rare, straight to memory)
            if (PC_WAVE_ANY(lw.ymin < lw.ymax)) {
                for (int32_t y = lw.ymin; y < lw.ymax; y++) {
                    const uint32_t t = lw.tiles_base + (uint32_t)(y - lw.bbox1) * (uint32_t)lw.stride;
                    if (t < cfg.tiles_size) atomicAdd(&tile[t].backdrop, lw.is_down ? -1 : 1);
                }
            }
        }
        if (lane == 0u) sh.wave_total[wave] = wave_total;
        __syncthreads();
        const uint32_t t0 = sh.wave_total[0], t1 = sh.wave_total[1], t2 = sh.wave_total[2], t3 = sh.wave_total[3];
        const uint32_t wave_off = (wave > 0u ? t0 : 0u) + (wave > 1u ? t1 : 0u) + (wave > 2u ? t2 : 0u);
        const uint32_t total = t0 + t1 + t2 + t3;
#ifdef VELLO_PC_TIMELINE
        const uint32_t tl1 = (uint32_t)wall_clock64();
#endif
        uint32_t reserved = 0u;
        if (tid == 0u && total) reserved = atomicAdd(&bump->seg_counts, total);  // (answers while the counting pass runs)
        // ---- counting pass ----
        uint32_t done = 0u;  // items of this wave before line group j (wave-uniform)
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) {
            pc_count_lines<P>(sh, w[j], w[j].ioff + p[j], done + p[j], n_stash[j], lane, wave, cfg, tile);
            done += T[j];
        }
        if (tid == 0u) sh.base = reserved;
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        const uint32_t tl2 = (uint32_t)wall_clock64();
#endif
        // ---- flush: one returning add per touched tile, 16 lanes on the 16 tiles of a cache line ----
        for (uint32_t k = tid; k < PC_TABLE; k += 256u)
            if (sh.keys[k] != PC_EMPTY) sh.occupied[atomicAdd(&sh.n_occ, 1u)] = (uint16_t)k;
        __syncthreads();
        const uint32_t n_occ = sh.n_occ;
        // PC_FLUSH_GROUP turns of the loop at a time: their returning adds are all requested before the first answer is waited
        // for.  (Round 6: written a turn at a time the loop was add, s_waitcnt vmcnt(0), ds_write -- a chunk's 200 occupied entries
        // are 12.5 turns of 256 lanes, 12.5 memory round trips in a row: 7 of a chunk's 20 us, profiles/r04_pc_timeline_agg_v5.txt.)
        for (uint32_t k0 = tid; k0 < n_occ * 16u; k0 += 256u * PC_FLUSH_GROUP) {
            uint32_t at[PC_FLUSH_GROUP], ix[PC_FLUSH_GROUP], word[PC_FLUSH_GROUP], old[PC_FLUSH_GROUP];
#pragma unroll
            for (uint32_t g = 0; g < PC_FLUSH_GROUP; g++) {
                const uint32_t k = k0 + g * 256u;
                const bool in = k < n_occ * 16u;
                const uint32_t e = in ? sh.occupied[k >> 4] : 0u, t = k & 15u;
                at[g] = e * 16u + t;
                word[g] = in ? sh.cnt[at[g]] : 0u;
                ix[g] = sh.keys[e] * 16u + t;
            }
#pragma unroll
            for (uint32_t g = 0; g < PC_FLUSH_GROUP; g++) {
                old[g] = 0u;
                if ((word[g] & 0xffffu) != 0u) old[g] = atomicAdd(&tile[ix[g]].segment_count_or_ix, word[g] & 0xffffu);
            }
#pragma unroll
            for (uint32_t g = 0; g < PC_FLUSH_GROUP; g++) {
                const int32_t d = (int32_t)word[g] >> 16;
                if (d != 0) atomicAdd(&tile[ix[g] + 1u].backdrop, d);  // (inside the pool: checked when the bump was added)
            }
#pragma unroll
            for (uint32_t g = 0; g < PC_FLUSH_GROUP; g++)
                if ((word[g] & 0xffffu) != 0u) sh.cnt[at[g]] = old[g];
        }
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        __builtin_amdgcn_s_waitcnt(0);
        const uint32_t tl4 = (uint32_t)wall_clock64();
#endif
        // ---- the records: a crossing per lane out of the stash ----
        const uint32_t seg_wave = sh.base + wave_off;
        done = 0u;
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) {
            for (uint32_t R = 0; R < V[j]; R += 64u) {
                const uint32_t q = R + lane;
                const bool act = q < V[j];
                const uint32_t word = act ? sh.stash[wave][done + q] : PC_DONE;
                const uint32_t owner = (word >> 16) & 63u;
                const uint32_t i = wave_shfl(w[j].ioff, owner) + q;
                if (act) {
                    uint32_t seg_within_slice = word & 0xffffu;
                    if (!(word & PC_DONE)) seg_within_slice = atomicAdd(&sh.cnt[word & 0xffffu], 1u);
                    const uint32_t seg_ix = seg_wave + done + q;
                    if (seg_ix < cfg.seg_counts_size) {
                        SegmentCount sc;
                        sc.line_ix = chunk + j * 256u + wave * 64u + owner;
                        sc.counts = (seg_within_slice << 16) | i;
                        seg_counts[seg_ix] = sc;
                    }
                }
            }
            if (PC_WAVE_ANY(n_stash[j] < w[j].count))  // crossings without a place in the stash: straight to memory
                pc_walk_line_direct(w[j], w[j].ioff + p[j], done + p[j], n_stash[j], cfg, tile, seg_wave, chunk + j * 256u + tid, seg_counts);
            done += T[j];
        }
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        const uint32_t tl5 = (uint32_t)wall_clock64();
#endif
        // ---- the table back to empty: only what the chunk touched ----
        for (uint32_t k = tid; k < n_occ * 16u; k += 256u) sh.cnt[sh.occupied[k >> 4] * 16u + (k & 15u)] = 0u;
        for (uint32_t k = tid; k < n_occ; k += 256u) sh.keys[sh.occupied[k]] = PC_EMPTY;
        if (tid == 0u) sh.n_occ = 0u;
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        if (tid == 0u) {
            const uint32_t slot = chunk / CHUNK;
            if (cfg.seg_counts_size > 4u * 8192u && slot < 8192u) {
                SegmentCount *dst = seg_counts + (cfg.seg_counts_size - 4u * 8192u) + 4u * slot;
                dst[0].line_ix = tl0; dst[0].counts = tl1;
                dst[1].line_ix = tl2; dst[1].counts = (uint32_t)wall_clock64();
                dst[2].line_ix = tl4; dst[2].counts = tl5;
                dst[3].line_ix = n_occ; dst[3].counts = 0u;
            }
        }
#endif
    }
}

// backdrop_dyn.wgsl:28-86: row-wise inclusive prefix of tile backdrops.
// The reference gives a row to a thread that walks it tile by tile (load -> add -> store, one memory latency per tile,
// lanes striding rows: nothing coalesces).  Here a WAVE owns a path and its lanes take 64 CONSECUTIVE tiles of the
// path's tile rectangle at a time -- rows are contiguous in it, so the loads and stores are whole cache lines -- with
// a segmented shuffle scan whose segments are the rows (a lane may add the lane d to its left iff its column is >= d)
// and a carry for the row that continues from the previous 64 tiles.  The road-map scene allocates 10 M path tiles.
// Rows are independent, so a path is cut into blocks of whole rows (about BACKDROP_BLOCK_TILES tiles) and the four
// waves of a workgroup share the blocks of FOUR consecutive paths (block b of the group's path p goes to wave
// (b + p) & 3): a typical path (99 tiles, one block) still gets a wave of its own, while the launch no longer ends with
// one wave's chain of 97 dependent steps through the largest road (6 k tiles).
constexpr uint32_t BACKDROP_BLOCK_TILES = 512u;  // (round 3 sweep on d2: 2048 -> 52 us, 1024 -> 48, 512 -> 44.5, 256 -> 44.5)
__global__ void __launch_bounds__(256) k_backdrop(Config cfg, const Bump *__restrict__ bump, const Path *__restrict__ paths, Tile *tiles) {
    if (bump->failed != 0u) return;
    const uint32_t lane = threadIdx.x & 63u, slice = threadIdx.x >> 6;
    const uint32_t n_obj = cfg.layout.n_draw_objects;
    for (uint32_t group = blockIdx.x; group * 4u < n_obj; group += gridDim.x) {
      // the group's four Path records are requested together (one round trip, not four in a row)
      Path group_paths[4];
#pragma unroll
      for (uint32_t p = 0; p < 4u; p++) group_paths[p] = load_path(paths, minu(group * 4u + p, n_obj - 1u));
#pragma unroll
      for (uint32_t p = 0; p < 4u; p++) {
        const uint32_t drawobj_ix = group * 4u + p;
        if (drawobj_ix >= n_obj) break;
        const Path path = group_paths[p];
        const uint32_t width = path.bbox[2] - path.bbox[0], height = path.bbox[3] - path.bbox[1];
        if (width <= 1u) continue;  // (a row of one tile is its own prefix)
        // (rows per block: any number >= 1 gives the same backdrops, and a scalar division is 35 instructions for each of the
        // four waves that look at the path -- the largest power of two with block_rows * width <= BACKDROP_BLOCK_TILES instead)
        const uint32_t block_rows = width >= BACKDROP_BLOCK_TILES ? 1u : BACKDROP_BLOCK_TILES >> (32u - (uint32_t)__builtin_clz(width - 1u));
        for (uint32_t row0 = ((slice - p) & 3u) * block_rows; row0 < height; row0 += 4u * block_rows) {
            const uint32_t first = path.tiles + row0 * width;
            const uint32_t n = minu(block_rows, height - row0) * width;
            int32_t carry = 0;
            // The tiles of the NEXT GROUP of four steps are requested before this group is scanned: a block is a chain of
            // steps, and since most steps are their load alone (below) what a wave has in flight is what it streams at --
            // one 512-byte request per wave was 2 TB/s over the pool (round 4: four).
            constexpr uint32_t G = 4u;
            int32_t next[G];
            // (every lane loads, from a clamped address, and what lies outside the block is zeroed afterwards: a load under a
            // branch makes the wait for THIS group's tiles a wait for everything in flight, the next group's request included
            // -- vmcnt counts in order and the compiler cannot count what a branch may have skipped)
            const uint32_t last_tile = cfg.tiles_size - 1u;
            auto request = [&](uint32_t base) {
#pragma unroll
#ifdef VK_BD_REQUEST_BEYOND_BLOCK  // (round 4's form: the requests run on into the next path's tiles)
                for (uint32_t k = 0; k < G; k++) next[k] = tiles[minu(first + base + k * 64u + lane, last_tile)].backdrop;
#else
                // (round 6: nothing is requested beyond the block's last tile -- the lanes past it load that tile again.  Round 4 kept the
                // requests running into the next path's tiles for 3 us one frame at a time; now the clamped form is 1.2 us faster alone
                // and +0.4 % with four frames in flight, profiles/r06_ab_backdrop_clamp.txt: 110 MB fetched for 80 MB of tiles before)
                for (uint32_t k = 0; k < G; k++) next[k] = tiles[minu(minu(first + base + k * 64u + lane, first + n - 1u), last_tile)].backdrop;
#endif
            };
            // one step: 64 tiles from tile i0 of the block on, their backdrops in `loaded`
            auto step = [&](uint32_t i0, int32_t loaded) {
                const uint32_t i = i0 + lane;
                const uint32_t tile_ix = first + i;
                const bool valid = i < n && tile_ix < cfg.tiles_size;
                int32_t v = valid ? loaded : 0;
                // Nothing to add up where nothing was bumped: 64 zero backdrops behind a zero carry are their own prefix
                // sums.  On a road map that is nearly every step (the two outlines of a stroke cancel within a tile or
                // two, and path_count never writes the cancelled pairs), and the step is then its load alone -- no column
                // arithmetic (an integer division), no six shuffle rounds.
                if (__ballot(v != 0) == 0ull && carry == 0) return;
                const uint32_t col = i % width;
                const int32_t own = v;
                // the rows' inclusive sums: a lane may add the lane d to its left iff its column is >= d.  (Round 6: inside the 16-lane
                // rows of the DPP by row shifts, then the totals of DPP rows 0 / 2 into rows 1 / 3 and of lane 31 into rows 2 and 3 where
                // the tile row began before them -- ten vector operations and no trip through the LDS crossbar; the six ds_bpermute
                // rounds, each with its wait, were most of the kernel's 9.6 M vector instructions: 60 % of the steps get here.)
#ifdef VELLO_SIMT_EMU
#pragma unroll
                for (uint32_t d = 1; d < 64u; d <<= 1) {
                    const int32_t up = __shfl_up(v, (int)d);
                    if (lane >= d && col >= d) v += up;
                }
#else
                {
                    uint32_t u = (uint32_t)v;
                    uint32_t t;
                    t = VK_DPP0(u, 0x111, 0xf); u += col >= 1u ? t : 0u;
                    t = VK_DPP0(u, 0x112, 0xf); u += col >= 2u ? t : 0u;
                    t = VK_DPP0(u, 0x114, 0xf); u += col >= 4u ? t : 0u;
                    t = VK_DPP0(u, 0x118, 0xf); u += col >= 8u ? t : 0u;
                    t = VK_DPP0(u, 0x142, 0xa); u += col > (lane & 15u) ? t : 0u;   // (rows 1, 3: lane 15 / 47; rows 0, 2 get 0)
                    t = VK_DPP0(u, 0x143, 0xc); u += col + 31u >= lane ? t : 0u;   // (rows 2, 3: lane 31; rows 0, 1 get 0)
                    v = (int32_t)u;
                }
#endif
                if (col > lane) v += carry;  // the row began before this step's first lane
                carry = __shfl(v, 63);
                if (valid && v != own) tiles[tile_ix].backdrop = v;
            };
            request(0u);
            for (uint32_t base = 0; base < n; base += 64u * G) {
                int32_t cur[G];
#pragma unroll
                for (uint32_t k = 0; k < G; k++) cur[k] = next[k];
                // (beyond the block: loads of the block's last tile that nobody reads, see `request`)
                request(base + 64u * G);
#pragma unroll
                for (uint32_t k = 0; k < G; k++) {
                    if (base + k * 64u >= n) break;
                    step(base + k * 64u, cur[k]);
                }
            }
        }
      }
    }
}

// path_tiling.wgsl:39-173: one (line, tile) crossing per thread, clipped to its tile.
__global__ void __launch_bounds__(256) k_path_tiling(Config cfg, Bump *bump, const SegmentCount *__restrict__ seg_counts,
                                                     const LineSoup *__restrict__ lines, const Path *__restrict__ paths,
                                                     const Tile *__restrict__ tiles, Segment *__restrict__ segments, uint32_t *ptcl) {
    if (bump->failed != 0u) {  // path_tiling_setup.wgsl:21-25
        if (blockIdx.x == 0 && threadIdx.x == 0) ptcl[0] = ~0u;
        return;
    }
    const float TILE_SCALE = 0.0625f;
    const uint32_t n_segments = minu(bump->seg_counts, cfg.seg_counts_size);
    for (uint32_t gi = blockIdx.x * 256u + threadIdx.x; gi < n_segments; gi += gridDim.x * 256u) {
        SegmentCount sc = seg_counts[gi];
        LineSoup line = lines[sc.line_ix];
        uint32_t seg_within_slice = sc.counts >> 16;
        uint32_t seg_within_line = sc.counts & 0xffffu;
        bool is_down = line.p1y >= line.p0y;
        vec2 xy0 = is_down ? v2(line.p0x, line.p0y) : v2(line.p1x, line.p1y);
        vec2 xy1 = is_down ? v2(line.p1x, line.p1y) : v2(line.p0x, line.p0y);
        vec2 s0 = xy0 * TILE_SCALE;
        vec2 s1 = xy1 * TILE_SCALE;
        uint32_t count_x = span(s0.x, s1.x) - 1u;
        uint32_t count = count_x + span(s0.y, s1.y);
        float dx = fabsf(s1.x - s0.x);
        float dy = s1.y - s0.y;
        float idxdy = 1.0f / (dx + dy);
        float a = dx * idxdy;
        bool is_positive_slope = s1.x >= s0.x;
        float x_sign = is_positive_slope ? 1.0f : -1.0f;
        float xt0 = floorf(s0.x * x_sign);
        float c = s0.x * x_sign - xt0;
        float y0i = floorf(s0.y);
        float ytop = (s0.y == s1.y) ? ceilf(s0.y) : y0i + 1.0f;
        float b = minf((dy * c + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
        float robust_err = floorf(a * ((float)count - 1.0f) + b) - (float)count_x;
        if (robust_err != 0.0f) a -= ROBUST_EPSILON * signf(robust_err);
        int32_t x0i = f2i(xt0 * x_sign + 0.5f * (x_sign - 1.0f));
        float z = floorf(a * (float)seg_within_line + b);
        int32_t x = x0i + f2i(x_sign * z);
        int32_t y = f2i(y0i + (float)seg_within_line - z);

        Path path = paths[line.path_ix];
        int32_t bbox0 = (int32_t)path.bbox[0], bbox1 = (int32_t)path.bbox[1], bbox2 = (int32_t)path.bbox[2];
        int32_t stride = bbox2 - bbox0;
        int32_t tile_ix = (int32_t)path.tiles + (y - bbox1) * stride + x - bbox0;
        // Tile indices are trusted upstream because WebGPU bounds every access; here the bound is explicit.  A line that
        // starts more than 65 535 tile crossings outside the viewport overflows the 16-bit crossing index of
        // SegmentCount (path_count.wgsl:196, the reference's own limit) and, like a crossing index beyond f32's 24 bits,
        // recomputes a tile that is not the path's: out of the buffer it reads as an empty tile, as a robust load would.
        Tile tile = (uint32_t)tile_ix < cfg.tiles_size ? tiles[tile_ix] : Tile{0, 0u};
        uint32_t seg_start = ~tile.segment_count_or_ix;
        if ((int32_t)seg_start < 0) continue;
        vec2 tile_xy = v2((float)x * (float)TILE_WIDTH, (float)y * (float)TILE_HEIGHT);
        vec2 tile_xy1 = v2(tile_xy.x + (float)TILE_WIDTH, tile_xy.y + (float)TILE_HEIGHT);
        if (seg_within_line > 0u) {
            float z_prev = floorf(a * ((float)seg_within_line - 1.0f) + b);
            if (z == z_prev) {
                float xt = xy0.x + (xy1.x - xy0.x) * (tile_xy.y - xy0.y) / (xy1.y - xy0.y);
                xt = clampf(xt, tile_xy.x + 1e-3f, tile_xy1.x);
                xy0 = v2(xt, tile_xy.y);
            } else {
                float x_clip = is_positive_slope ? tile_xy.x : tile_xy1.x;
                float yt = xy0.y + (xy1.y - xy0.y) * (x_clip - xy0.x) / (xy1.x - xy0.x);
                yt = clampf(yt, tile_xy.y + 1e-3f, tile_xy1.y);
                xy0 = v2(x_clip, yt);
            }
        }
        if (seg_within_line < count - 1u) {
            float z_next = floorf(a * ((float)seg_within_line + 1.0f) + b);
            if (z == z_next) {
                float xt = xy0.x + (xy1.x - xy0.x) * (tile_xy1.y - xy0.y) / (xy1.y - xy0.y);
                xt = clampf(xt, tile_xy.x + 1e-3f, tile_xy1.x);
                xy1 = v2(xt, tile_xy1.y);
            } else {
                float x_clip = is_positive_slope ? tile_xy1.x : tile_xy.x;
                float yt = xy0.y + (xy1.y - xy0.y) * (x_clip - xy0.x) / (xy1.x - xy0.x);
                yt = clampf(yt, tile_xy.y + 1e-3f, tile_xy1.y);
                xy1 = v2(x_clip, yt);
            }
        }
        float y_edge = 1e9f;
        vec2 p0 = xy0 - tile_xy;
        vec2 p1 = xy1 - tile_xy;
        const float EPSILON = 1e-6f;
        if (p0.x == 0.0f) {
            if (p1.x == 0.0f) {
                p0.x = EPSILON;
                if (p0.y == 0.0f) {
                    p1.x = EPSILON;
                    p1.y = (float)TILE_HEIGHT;
                } else {
                    p1.x = 2.0f * EPSILON;
                    p1.y = p0.y;
                }
            } else if (p0.y == 0.0f) {
                p0.x = EPSILON;
            } else {
                y_edge = p0.y;
            }
        } else if (p1.x == 0.0f) {
            if (p1.y == 0.0f) {
                p1.x = EPSILON;
            } else {
                y_edge = p1.y;
            }
        }
        if (p0.x == floorf(p0.x) && p0.x != 0.0f) p0.x -= EPSILON;
        if (p1.x == floorf(p1.x) && p1.x != 0.0f) p1.x -= EPSILON;
        if (!is_down) {
            vec2 tmp = p0;
            p0 = p1;
            p1 = tmp;
        }
        uint32_t out_ix = seg_start + seg_within_slice;
        if (out_ix < cfg.segments_size) {
            Segment sg;
            sg.p0x = p0.x; sg.p0y = p0.y; sg.p1x = p1.x; sg.p1y = p1.y;
            sg.y_edge = y_edge;
            sg.pad = 0u;
            segments[out_ix] = sg;
        }
    }
}

static uint32_t clamp_grid(uint64_t work_items, uint32_t per_block, uint32_t max_blocks) {
    uint64_t g = (work_items + per_block - 1u) / per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (uint32_t)g;
}

void launch_path_count(const Frame &f, hipStream_t s) {
    if (f.path_count_small) {
        const uint32_t grid = clamp_grid(f.cfg.lines_size, 256u, VK_PC_GRID);
        hipLaunchKernelGGL((k_path_count<1u, PcAlone>), dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.lines, f.paths, f.tiles, f.seg_counts);
    } else if (f.flatten_side_by_side) {  // (one frame in flight)
        const uint32_t grid = clamp_grid(f.cfg.lines_size, PATH_COUNT_CHUNK, VK_PC_GRID);
        hipLaunchKernelGGL((k_path_count<PATH_COUNT_LINES_PER_THREAD, PcAlone>), dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.lines, f.paths, f.tiles, f.seg_counts);
    } else {  // frames in flight: the small footprint (PcInFlight, above)
        const uint32_t grid = clamp_grid(f.cfg.lines_size, 256u * VK_PC_LPT_IN_FLIGHT, VK_PC_GRID_IN_FLIGHT);
        hipLaunchKernelGGL((k_path_count<VK_PC_LPT_IN_FLIGHT, PcInFlight>), dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.lines, f.paths, f.tiles, f.seg_counts);
    }
}

void launch_backdrop(const Frame &f, hipStream_t s) {
    if (f.cfg.layout.n_paths == 0) return;
    // a workgroup per four paths (its waves share their row blocks), workgroups striding over the groups
    uint32_t n_wg = (f.cfg.layout.n_paths + 3u) / 4u;
#ifndef VK_BD_GRID_IN_FLIGHT
#define VK_BD_GRID_IN_FLIGHT 8192u  // (sweep constant: the grid's cap with frames in flight)
#endif
    const uint32_t cap = f.flatten_side_by_side ? 8192u : VK_BD_GRID_IN_FLIGHT;
    if (n_wg > cap) n_wg = cap;
    hipLaunchKernelGGL(k_backdrop, dim3(n_wg), dim3(256), 0, s, f.cfg, f.bump(), f.paths, f.tiles);
}

void launch_path_tiling(const Frame &f, hipStream_t s) {
#ifndef VK_PT_GRID_IN_FLIGHT
#define VK_PT_GRID_IN_FLIGHT 2048u  // (sweep constant: the grid's cap with frames in flight)
#endif
    uint32_t grid = clamp_grid(f.cfg.seg_counts_size, 256u, f.flatten_side_by_side ? 2048u : VK_PT_GRID_IN_FLIGHT);
    hipLaunchKernelGGL(k_path_tiling, dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.seg_counts, f.lines, f.paths, f.tiles,
                       f.segments, f.ptcl);
}

}  // namespace vk
