// path_count (+setup), backdrop_dyn, path_tiling (+setup).
// Reference: vello_shaders/shader/path_count_setup.wgsl:17-27, path_count.wgsl:51-202,
// backdrop_dyn.wgsl:28-86, path_tiling_setup.wgsl:20-32, path_tiling.wgsl:39-173
// (vello/src/render.rs:443-502); CPU twins cpu/{path_count,backdrop,path_tiling}.rs.
//
// HIP has no indirect dispatch: both line-driven kernels launch a fixed grid and stride over the
// count the previous stage left in `bump` (the *_setup dispatches disappear).
#include "engine.h"

#ifndef VK_PC_AGG
#define VK_PC_AGG 0
#endif

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
template <class GetPath>
__device__ __forceinline__ LineWalk setup_line_walk(const LineSoup &line, GetPath get_path, uint32_t n_paths) {
    LineWalk w = {};
    // A tag stream with more PATH markers than the layout counts (only a hand-made stream: resolve appends its extra
    // markers behind the last segment, resolve.rs:127-129) yields lines whose path has no Path record.  WebGPU reads
    // zeros there (stride 0 -> no crossings, path_count.wgsl:112); HIP would read past paths[].
    if (line.path_ix >= n_paths) return w;
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
    if (dx + dy == 0.0f) return w;
    if (dy == 0.0f && floorf(s0.y) == s0.y) return w;
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
    if (s0.y >= (float)bbox3 || s1.y <= (float)bbox1 || xmin >= (float)bbox2 || stride == 0) return w;
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
    w.valid = true;
    w.is_down = is_down;
    w.is_positive_slope = is_positive_slope;
    w.a = a; w.b = b; w.x0 = x0; w.y0 = y0; w.x_sign = x_sign; w.s0x = s0.x; w.s0y = s0.y;
    w.imin = imin; w.imax = imax;
    w.ymin = maxi(ymin, bbox1);
    w.ymax = mini(ymax, bbox3);
    w.bbox0 = bbox0; w.bbox1 = bbox1; w.bbox2 = bbox2; w.bbox3 = bbox3; w.stride = stride;
    w.tiles_base = path.tiles;
    return w;
}

}  // namespace

// One workgroup handles chunks of 256 x 8 lines.  Pass 1 derives every line's crossing count
// (imax - imin), a shuffle scan + ONE atomicAdd(bump.seg_counts) reserves the chunk's slice, pass 2
// re-derives the walk (cheap, line and path hit L2) and writes backdrops, per-tile counts and the
// SegmentCount records.  The reference issues one bump atomic per line (path_count.wgsl:172).
// KEEP: pass 1 sets the lines and Path records aside in LDS for pass 2 (36 KB: four workgroups per CU instead of six); the
// host picks the form by the size of the soup (engine.h PATH_COUNT_KEEP_MIN_LINES).
template <bool KEEP>
__global__ void __launch_bounds__(256) k_path_count(Config cfg, Bump *bump, const LineSoup *__restrict__ lines,
                                                    const Path *__restrict__ paths, Tile *tile, SegmentCount *__restrict__ seg_counts) {
    __shared__ uint32_t sh_scan[4];
    __shared__ uint32_t sh_base;
    // A thread's four lines and their Path records, set aside by pass 1 for pass 2 (each thread reads back what it wrote: no
    // barrier).  Reloading them cost pass 2 24 of its 57 us per chunk on the road map -- a dependent pair of loads per round
    // at the 3 us a load takes while every workgroup's tile atomics are in flight (thread 0's stamps, scripts/pc_timeline.py).
    __shared__ uint32_t sh_keep[9][KEEP ? PATH_COUNT_CHUNK : 1u];
    const uint32_t tid = threadIdx.x;
    if (bump->failed != 0u) return;  // path_count_setup.wgsl:18-19
    const uint32_t n_lines = minu(bump->lines, cfg.lines_size);
    for (uint32_t chunk = blockIdx.x * PATH_COUNT_CHUNK; chunk < n_lines; chunk += gridDim.x * PATH_COUNT_CHUNK) {
#ifdef VELLO_PC_TIMELINE
        // measurement build (scripts/pc_timeline.py): per chunk, wall-clock stamps (100 MHz) of start / pass 1 done / slots
        // reserved / pass 2 done in the tail of the SegmentCount pool
        const uint32_t tl0 = (uint32_t)wall_clock64();
        uint32_t tl1 = 0u, tl2 = 0u;
        // ... and inside pass 2, as thread 0 sees them: ticks in the rounds' setup (line + Path loads, the walk's parameters), in
        // the row loops of lines left of the rectangle, in the lockstep rounds (atomics + records); rounds of four steps walked
        uint32_t tl_setup = 0u, tl_rows = 0u, tl_walk = 0u, tl_rounds = 0u;
#endif
        uint32_t my_total = 0u;
#pragma unroll 1
        for (uint32_t j = 0; j < PATH_COUNT_LINES_PER_THREAD; j++) {
            uint32_t line_ix = chunk + j * 256u + tid;
            if (line_ix < n_lines) {
                const LineSoup line = load_line(lines, line_ix);
                const uint32_t at = KEEP ? j * 256u + tid : 0u;
                if (KEEP) {
                    // (a line whose path has no record is set aside as a point: no crossings, and no Path asked for, in pass 2 as here)
                    const bool known = line.path_ix < cfg.layout.n_paths;
                    sh_keep[1][at] = known ? __float_as_uint(line.p0x) : 0u; sh_keep[2][at] = known ? __float_as_uint(line.p0y) : 0u;
                    sh_keep[3][at] = known ? __float_as_uint(line.p1x) : 0u; sh_keep[4][at] = known ? __float_as_uint(line.p1y) : 0u;
                }
                LineWalk w = setup_line_walk(line, [&]() {
                    const Path p = load_path(paths, line.path_ix);
                    if (KEEP) {
                        sh_keep[5][at] = p.bbox[0]; sh_keep[6][at] = p.bbox[1]; sh_keep[7][at] = p.bbox[2]; sh_keep[8][at] = p.bbox[3];
                        sh_keep[0][at] = p.tiles;
                    }
                    return p;
                }, cfg.layout.n_paths);
                my_total += w.imax - w.imin;
            }
        }
        uint32_t total;
        uint32_t incl = block256_incl_scan_u32(my_total, sh_scan, &total);
#ifdef VELLO_PC_TIMELINE
        tl1 = (uint32_t)wall_clock64();
#endif
        if (tid == 0u) sh_base = total ? atomicAdd(&bump->seg_counts, total) : 0u;
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        tl2 = (uint32_t)wall_clock64();
#endif
        uint32_t seg_base = sh_base + (incl - my_total);
        const int lane = (int)(tid & 63u);
#pragma unroll 1
        for (uint32_t j = 0; j < PATH_COUNT_LINES_PER_THREAD; j++) {
            uint32_t line_ix = chunk + j * 256u + tid;
#ifdef VELLO_PC_TIMELINE
            const uint32_t tj0 = (uint32_t)wall_clock64();
#endif
            LineWalk w = {};
            if (!KEEP) {
                if (line_ix < n_lines) {
                    const LineSoup line = load_line(lines, line_ix);
                    w = setup_line_walk(line, [&]() { return load_path(paths, line.path_ix); }, cfg.layout.n_paths);
                }
            } else if (line_ix < n_lines) {
                const uint32_t at = j * 256u + tid;
                LineSoup line;
                line.path_ix = 0u; line.pad = 0u;  // (checked against n_paths by pass 1)
                line.p0x = __uint_as_float(sh_keep[1][at]); line.p0y = __uint_as_float(sh_keep[2][at]);
                line.p1x = __uint_as_float(sh_keep[3][at]); line.p1y = __uint_as_float(sh_keep[4][at]);
                w = setup_line_walk(line, [&]() {
                    Path p;
                    p.bbox[0] = sh_keep[5][at]; p.bbox[1] = sh_keep[6][at]; p.bbox[2] = sh_keep[7][at]; p.bbox[3] = sh_keep[8][at];
                    p.tiles = sh_keep[0][at];
                    p.pad[0] = p.pad[1] = p.pad[2] = 0u;
                    return p;
                }, 1u);
            }
            uint32_t count = w.valid ? w.imax - w.imin : 0u;
            const int32_t delta = w.is_down ? -1 : 1;
#ifdef VELLO_PC_TIMELINE
            count = opaque(count);  // (the stamp waits for the loads the count depends on)
            const uint32_t tj1 = (uint32_t)wall_clock64();
            tl_setup += tj1 - tj0;
#endif
            // every tile index below is bounded by the buffer explicitly (WebGPU does that for the reference): with
            // crossing indices past f32's 24 bits the walk can leave the path's tile rectangle
            if (w.valid) {
                for (int32_t y = w.ymin; y < w.ymax; y++) {
                    int32_t base = (int32_t)w.tiles_base + (y - w.bbox1) * w.stride;
                    if ((uint32_t)base < cfg.tiles_size) atomicAdd(&tile[base].backdrop, delta);
                }
            }
            float last_z = floorf(w.a * ((float)w.imin - 1.0f) + w.b);
#ifdef VELLO_PC_TIMELINE
            const uint32_t tj2 = (uint32_t)wall_clock64();
            tl_rows += tj2 - tj1;
#endif
            // The wave walks crossings in lockstep.  Consecutive lanes hold consecutive lines of the soup, which
            // (flatten writes in tag order) are consecutive short segments of one path and mostly fall into the
            // same tile: runs of adjacent lanes hitting the same tile reserve their slots with ONE returning
            // atomic issued by the run head (the reference does one per crossing, path_count.wgsl:189).
            uint32_t max_count = count;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) max_count = maxu(max_count, __shfl_xor(max_count, d));
            // Four crossing steps per round: their four returning atomics are in flight together, the records are
            // written once all have answered (one atomic round trip per step made long lines -- tiger, mmark --
            // pay ~1.5 us per crossed tile).
            for (uint32_t s0 = 0; s0 < max_count; s0 += 4u) {
                bool k_act[4];
                uint32_t k_i[4], k_r[4];
                int k_head[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) {
                    const uint32_t s = s0 + k;
                    const bool act = s < count;
                    uint32_t key = 0xffffffffu, bkey = 0xffffffffu;
                    uint32_t i = w.imin + s;
                    if (act) {
                        float zf = w.a * (float)i + w.b;
                        float z = floorf(zf);
                        int32_t y = f2i(w.y0 + (float)i - z);
                        int32_t x = f2i(w.x0 + w.x_sign * z);
                        int32_t base = (int32_t)w.tiles_base + (y - w.bbox1) * w.stride - w.bbox0;
                        bool top_edge = (i == 0u) ? (w.y0 == w.s0y) : (last_z == z);
                        if (top_edge && x + 1 < w.bbox2) {
                            int32_t x_bump = maxi(x + 1, w.bbox0);
                            if ((uint32_t)(base + x_bump) < cfg.tiles_size) bkey = (uint32_t)(base + x_bump);
                        }
                        key = (uint32_t)(base + x);
                        last_z = z;
                    }
                    // Backdrop bumps of ADJACENT lanes on the same tile are added up before they touch memory, in pairs
                    // (even offsets of a run of equal targets absorb their right neighbour).  The two outlines of a stroked
                    // segment are neighbours in the soup (flatten allocates them together), run a few pixels apart in
                    // opposite directions and cross a tile row in the same tile: their +1 and -1 cancel and no atomic is
                    // issued at all.  What a scattered atomic costs is one request per distinct cache line and
                    // instruction, 2.7e10 per second chip-wide (scripts/calib/atomic_rate.hip) -- that rate, not
                    // arithmetic or bandwidth, is what bounds this kernel.  Integer adds: any grouping gives the same sum.
                    if (__ballot(bkey != 0xffffffffu) != 0ull) {
                        const uint32_t bprev = __shfl_up(bkey, 1), bnext = __shfl_down(bkey, 1);
                        const int32_t dnext = __shfl_down(delta, 1);
                        const unsigned long long bheads = __ballot(lane == 0 || bprev != bkey);
                        const int bhead_lane = 63 - __clzll((long long)(bheads & (~0ull >> (63 - lane))));
                        if (bkey != 0xffffffffu && (((lane - bhead_lane) & 1) == 0)) {
                            const int32_t d = delta + ((lane < 63 && bnext == bkey) ? dnext : 0);
                            if (d != 0) atomicAdd(&tile[bkey].backdrop, d);
                        }
                    }
                    uint32_t prev_key = __shfl_up(key, 1);
                    bool head = !act || lane == 0 || prev_key != key;
                    unsigned long long heads = __ballot(head);
                    unsigned long long le = heads & (~0ull >> (63 - lane));       // heads at lanes <= mine
                    int head_lane = 63 - __clzll((long long)le);
                    unsigned long long gt = lane == 63 ? 0ull : (heads & (~0ull << (lane + 1)));  // heads after me
                    int run_end = gt ? (__ffsll((long long)gt) - 1) : 64;
                    uint32_t r = 0u;
                    if (act && head && key < cfg.tiles_size) r = atomicAdd(&tile[key].segment_count_or_ix, (uint32_t)(run_end - lane));
                    k_act[k] = act;
                    k_i[k] = i;
                    k_r[k] = r;
                    k_head[k] = head_lane;
                }
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) {
                    uint32_t base_slot = __shfl(k_r[k], k_head[k]);
                    if (k_act[k]) {
                        uint32_t seg_within_slice = base_slot + (uint32_t)(lane - k_head[k]);
                        uint32_t seg_ix = seg_base + s0 + k;
                        if (seg_ix < cfg.seg_counts_size) {
                            SegmentCount sc;
                            sc.line_ix = line_ix;
                            sc.counts = (seg_within_slice << 16) | k_i[k];
                            seg_counts[seg_ix] = sc;
                        }
                    }
                }
            }
            seg_base += count;
#ifdef VELLO_PC_TIMELINE
            seg_base = opaque(seg_base);
            __builtin_amdgcn_s_waitcnt(0);  // (the round's atomics have answered, its records are on their way)
            tl_walk += (uint32_t)wall_clock64() - tj2;
            tl_rounds += (max_count + 3u) / 4u;
#endif
        }
        __syncthreads();  // sh_base / sh_scan reuse in the next chunk
#ifdef VELLO_PC_TIMELINE
        if (tid == 0u) {
            const uint32_t slot = chunk / PATH_COUNT_CHUNK;
            if (cfg.seg_counts_size > 4u * 8192u && slot < 8192u) {
                SegmentCount *dst = seg_counts + (cfg.seg_counts_size - 4u * 8192u) + 4u * slot;
                dst[0].line_ix = tl0; dst[0].counts = tl1;
                dst[1].line_ix = tl2; dst[1].counts = (uint32_t)wall_clock64();
                dst[2].line_ix = tl_setup; dst[2].counts = tl_rows;
                dst[3].line_ix = tl_walk; dst[3].counts = tl_rounds;
            }
        }
#endif
    }
}

// ---- k_path_count_agg: the tile atomics of a whole workgroup chunk added up in LDS first -------------------------------
//
// What a scattered atomic costs is one request per (instruction, distinct cache line), 2.7e10 per second chip-wide, whether
// the lines come from one XCD or from all (scripts/calib/atomic_rate.hip, atomic_scope.hip); k_path_count above issues 2.0-2.3
// M of them on the road map (scripts/pc_requests.py) and runs at that rate.  The 1 024 lines of a chunk are consecutive lines
// of a few paths and cross the same tiles over and over: per chunk 1 430 crossings fall into 475 tiles on 207 cache lines of
// the tile pool.  So the workgroup counts in LDS -- a small hash table keyed by the cache line (16 tiles), a word per tile:
// crossings in the low half, the sum of the top-edge backdrop bumps in the high half -- and then asks memory ONCE per touched
// tile, 16 lanes on the 16 tiles of a line: 0.7-0.9 M requests.  The returned old value of a tile becomes the cursor its
// crossings draw their slot index from (a returning LDS add) when the walk is repeated to write the SegmentCount records.
// Which crossing of a tile gets which slot is as arbitrary as in the reference (path_count.wgsl:189 is an atomicAdd in
// dispatch order).  A line whose cache line finds no place in the table (PC_PROBES slots taken by other lines: the table
// only grows within a chunk, so both walks see the same answer) goes to memory directly, as every crossing used to.
// The walks of a thread's lines live in registers between the passes (13 words each).
#ifndef VK_PC_TABLE_LOG2
#define VK_PC_TABLE_LOG2 9
#endif
constexpr uint32_t PC_TABLE_LOG2 = VK_PC_TABLE_LOG2, PC_TABLE = 1u << PC_TABLE_LOG2, PC_PROBES = 4u;
constexpr uint32_t PC_EMPTY = 0xffffffffu, PC_NONE = 0xffffffffu;

struct PcTable {
    uint32_t keys[PC_TABLE];     // tile index >> 4 of the entry, PC_EMPTY
    uint32_t cnt[PC_TABLE][16];  // pass A: crossings | backdrop sum << 16; after the flush: the tile's slot cursor
    uint16_t occupied[PC_TABLE];
    uint32_t n_occ;
};

struct PcWalk {
    float a, b, x0, y0;
    uint32_t imin, count;
    int32_t ymin, ymax;          // rows bumped in column 0 by a line left of the rectangle
    uint32_t base;               // tile index of (x, y) = base + y * stride + x
    int32_t stride, bbox0, bbox2;
    uint32_t flags;              // 1: is_down, 2: negative slope, 4: y0 == s0.y
};

__device__ __forceinline__ PcWalk pc_pack(const LineWalk &w) {
    PcWalk p;
    p.a = w.a; p.b = w.b; p.x0 = w.x0; p.y0 = w.y0;
    p.imin = w.imin;
    p.count = w.valid ? w.imax - w.imin : 0u;
    p.ymin = w.valid ? w.ymin : 0; p.ymax = w.valid ? w.ymax : 0;
    p.base = w.tiles_base - (uint32_t)w.bbox1 * (uint32_t)w.stride - (uint32_t)w.bbox0;
    p.stride = w.stride; p.bbox0 = w.bbox0; p.bbox2 = w.bbox2;
    p.flags = (w.is_down ? 1u : 0u) | (w.is_positive_slope ? 0u : 2u) | (w.y0 == w.s0y ? 4u : 0u);
    return p;
}

// crossing s of a walk: its tile and the tile its top edge bumps (PC_NONE: none, or outside the pool)
__device__ __forceinline__ void pc_crossing(const PcWalk &w, uint32_t s, float &last_z, uint32_t tiles_size, uint32_t &key, uint32_t &bkey, uint32_t &i_out) {
    const uint32_t i = w.imin + s;
    const float z = floorf(w.a * (float)i + w.b);
    const int32_t y = f2i(w.y0 + (float)i - z);
    const int32_t x = f2i(w.x0 + ((w.flags & 2u) ? -1.0f : 1.0f) * z);
    const uint32_t row = w.base + (uint32_t)y * (uint32_t)w.stride;
    const bool top_edge = (i == 0u) ? ((w.flags & 4u) != 0u) : (last_z == z);
    bkey = PC_NONE;
    if (top_edge && x + 1 < w.bbox2) {
        const uint32_t t = row + (uint32_t)maxi(x + 1, w.bbox0);
        if (t < tiles_size) bkey = t;
    }
    key = row + (uint32_t)x;
    last_z = z;
    i_out = i;
}

__device__ __forceinline__ uint32_t pc_hash(uint32_t line) { return (line * 0x9E3779B1u) >> (32u - PC_TABLE_LOG2); }

template <bool INSERT>
__device__ __forceinline__ uint32_t pc_slot(PcTable &t, uint32_t line) {
    uint32_t s = pc_hash(line);
#pragma unroll 1
    for (uint32_t p = 0; p < PC_PROBES; p++) {
        uint32_t k = __atomic_load_n(&t.keys[s], __ATOMIC_RELAXED);
        if (k == line) return s;
        if (k == PC_EMPTY) {
            if (!INSERT) return PC_NONE;
            k = atomicCAS(&t.keys[s], PC_EMPTY, line);
            if (k == PC_EMPTY) {
                t.occupied[atomicAdd(&t.n_occ, 1u)] = (uint16_t)s;
                return s;
            }
            if (k == line) return s;
        }
        s = (s + 1u) & (PC_TABLE - 1u);
    }
    return PC_NONE;
}

// the slot of a tile's cache line, remembering the last answer (a walk stays on a line for several steps)
template <bool INSERT>
__device__ __forceinline__ uint32_t pc_slot_of(PcTable &t, uint32_t tile_ix, uint32_t &last_line, uint32_t &last_slot) {
    const uint32_t line = tile_ix >> 4;
    if (line != last_line) {
        last_line = line;
        last_slot = pc_slot<INSERT>(t, line);
    }
    return last_slot;
}

template <uint32_t LPT>
__global__ void __launch_bounds__(256) k_path_count_agg(Config cfg, Bump *bump, const LineSoup *__restrict__ lines,
                                                        const Path *__restrict__ paths, Tile *tile, SegmentCount *__restrict__ seg_counts) {
    __shared__ uint32_t sh_scan[4];
    __shared__ uint32_t sh_base;
    __shared__ PcTable tb;
#ifdef VELLO_PC_TIMELINE
    __shared__ uint32_t sh_direct;  // measurement build: crossings and bumps that found no place in the table
    if (threadIdx.x == 0u) sh_direct = 0u;
#endif
    const uint32_t tid = threadIdx.x;
    if (bump->failed != 0u) return;  // path_count_setup.wgsl:18-19
    const uint32_t n_lines = minu(bump->lines, cfg.lines_size);
    const uint32_t n_paths = cfg.layout.n_paths;
    for (uint32_t k = tid; k < PC_TABLE; k += 256u) tb.keys[k] = PC_EMPTY;
    for (uint32_t k = tid; k < PC_TABLE * 16u; k += 256u) (&tb.cnt[0][0])[k] = 0u;
    if (tid == 0u) tb.n_occ = 0u;
    __syncthreads();
    constexpr uint32_t CHUNK = 256u * LPT;
    for (uint32_t chunk = blockIdx.x * CHUNK; chunk < n_lines; chunk += gridDim.x * CHUNK) {
#ifdef VELLO_PC_TIMELINE
        // measurement build (scripts/pc_timeline.py): wall-clock stamps (100 MHz) of the chunk's phases in the tail of the pool
        const uint32_t tl0 = (uint32_t)wall_clock64();
#endif
        // ---- pass 1: the walks; the chunk's slice of the SegmentCount pool ----
        LineSoup ln[LPT];
        Path pa[LPT];
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) ln[j] = load_line(lines, minu(chunk + j * 256u + tid, n_lines - 1u));
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) pa[j] = load_path(paths, ln[j].path_ix < n_paths ? ln[j].path_ix : 0u);  // (the pool holds >= 256 records)
        PcWalk w[LPT];
        uint32_t my_total = 0u;
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) {
            LineWalk lw = {};
            if (chunk + j * 256u + tid < n_lines) lw = setup_line_walk(ln[j], [&]() { return pa[j]; }, n_paths);
            w[j] = pc_pack(lw);
            my_total += w[j].count;
        }
        uint32_t total;
        const uint32_t incl = block256_incl_scan_u32(my_total, sh_scan, &total);
#ifdef VELLO_PC_TIMELINE
        const uint32_t tl1 = (uint32_t)wall_clock64();
#endif
        uint32_t reserved = 0u;
        if (tid == 0u && total) reserved = atomicAdd(&bump->seg_counts, total);  // (answers while pass A runs)
        // ---- pass A: count into the table ----
        uint32_t last_line = PC_NONE, last_slot = PC_NONE;
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) {
            const uint32_t dword = (w[j].flags & 1u) ? 0xffff0000u : 0x00010000u;  // -1 or +1 in the high half
            for (int32_t y = w[j].ymin; y < w[j].ymax; y++) {
                const uint32_t t = w[j].base + (uint32_t)w[j].bbox0 + (uint32_t)y * (uint32_t)w[j].stride;
                if (t < cfg.tiles_size) {
                    const uint32_t slot = pc_slot_of<true>(tb, t, last_line, last_slot);
                    if (slot != PC_NONE) atomicAdd(&tb.cnt[slot][t & 15u], dword);
                }
            }
            float last_z = floorf(w[j].a * ((float)w[j].imin - 1.0f) + w[j].b);
            for (uint32_t s = 0; s < w[j].count; s++) {
                uint32_t key, bkey, i;
                pc_crossing(w[j], s, last_z, cfg.tiles_size, key, bkey, i);
                if (key < cfg.tiles_size) {
                    const uint32_t slot = pc_slot_of<true>(tb, key, last_line, last_slot);
                    if (slot != PC_NONE) atomicAdd(&tb.cnt[slot][key & 15u], 1u);
                }
                if (bkey != PC_NONE) {
                    const uint32_t slot = pc_slot_of<true>(tb, bkey, last_line, last_slot);
                    if (slot != PC_NONE) atomicAdd(&tb.cnt[slot][bkey & 15u], dword);
                }
            }
        }
        if (tid == 0u) sh_base = reserved;
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        const uint32_t tl2 = (uint32_t)wall_clock64();
#endif
        // ---- flush: one returning add per touched tile, 16 lanes on the 16 tiles of a cache line ----
        const uint32_t n_occ = tb.n_occ;
        for (uint32_t k = tid; k < n_occ * 16u; k += 256u) {
            const uint32_t e = tb.occupied[k >> 4], t = k & 15u;
            const uint32_t word = tb.cnt[e][t];
            const uint32_t n = word & 0xffffu;
            const int32_t d = (int32_t)word >> 16;
            const uint32_t ix = tb.keys[e] * 16u + t;
            if (n != 0u) tb.cnt[e][t] = atomicAdd(&tile[ix].segment_count_or_ix, n);
            if (d != 0) atomicAdd(&tile[ix].backdrop, d);
        }
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        __builtin_amdgcn_s_waitcnt(0);
        const uint32_t tl4 = (uint32_t)wall_clock64();
#endif
        // ---- pass B: the records ----
        uint32_t seg_base = sh_base + (incl - my_total);
        last_line = PC_NONE; last_slot = PC_NONE;
#pragma unroll
        for (uint32_t j = 0; j < LPT; j++) {
            const uint32_t line_ix = chunk + j * 256u + tid;
            const int32_t delta = (w[j].flags & 1u) ? -1 : 1;
            for (int32_t y = w[j].ymin; y < w[j].ymax; y++) {
                const uint32_t t = w[j].base + (uint32_t)w[j].bbox0 + (uint32_t)y * (uint32_t)w[j].stride;
                if (t < cfg.tiles_size && pc_slot_of<false>(tb, t, last_line, last_slot) == PC_NONE) atomicAdd(&tile[t].backdrop, delta);
            }
            float last_z = floorf(w[j].a * ((float)w[j].imin - 1.0f) + w[j].b);
            for (uint32_t s = 0; s < w[j].count; s++) {
                uint32_t key, bkey, i;
                pc_crossing(w[j], s, last_z, cfg.tiles_size, key, bkey, i);
                uint32_t seg_within_slice = 0u;
                if (key < cfg.tiles_size) {
                    const uint32_t slot = pc_slot_of<false>(tb, key, last_line, last_slot);
                    if (slot != PC_NONE) seg_within_slice = atomicAdd(&tb.cnt[slot][key & 15u], 1u);
                    else {
                        seg_within_slice = atomicAdd(&tile[key].segment_count_or_ix, 1u);
#ifdef VELLO_PC_TIMELINE
                        atomicAdd(&sh_direct, 1u);
#endif
                    }
                }
                if (bkey != PC_NONE && pc_slot_of<false>(tb, bkey, last_line, last_slot) == PC_NONE) atomicAdd(&tile[bkey].backdrop, delta);
                const uint32_t seg_ix = seg_base + s;
                if (seg_ix < cfg.seg_counts_size) {
                    SegmentCount sc;
                    sc.line_ix = line_ix;
                    sc.counts = (seg_within_slice << 16) | i;
                    seg_counts[seg_ix] = sc;
                }
            }
            seg_base += w[j].count;
        }
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        const uint32_t tl5 = (uint32_t)wall_clock64();
#endif
        // ---- the table back to empty: only what the chunk touched ----
        for (uint32_t k = tid; k < n_occ * 16u; k += 256u) tb.cnt[tb.occupied[k >> 4]][k & 15u] = 0u;
        for (uint32_t k = tid; k < n_occ; k += 256u) tb.keys[tb.occupied[k]] = PC_EMPTY;
        if (tid == 0u) tb.n_occ = 0u;
        __syncthreads();
#ifdef VELLO_PC_TIMELINE
        if (tid == 0u) {
            const uint32_t slot = chunk / CHUNK;
            if (cfg.seg_counts_size > 4u * 8192u && slot < 8192u) {
                SegmentCount *dst = seg_counts + (cfg.seg_counts_size - 4u * 8192u) + 4u * slot;
                dst[0].line_ix = tl0; dst[0].counts = tl1;
                dst[1].line_ix = tl2; dst[1].counts = (uint32_t)wall_clock64();
                dst[2].line_ix = tl4; dst[2].counts = tl5;
                dst[3].line_ix = n_occ; dst[3].counts = sh_direct;
                sh_direct = 0u;
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
        const uint32_t block_rows = maxu(1u, BACKDROP_BLOCK_TILES / width);
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
                for (uint32_t k = 0; k < G; k++) next[k] = tiles[minu(first + base + k * 64u + lane, last_tile)].backdrop;
            };
            request(0u);
            for (uint32_t base = 0; base < n; base += 64u * G) {
                int32_t cur[G];
#pragma unroll
                for (uint32_t k = 0; k < G; k++) cur[k] = next[k];
                request(base + 64u * G);  // (beyond the block: clamped loads nobody reads)
#pragma unroll
                for (uint32_t k = 0; k < G; k++) {
                    const uint32_t i = base + k * 64u + lane;
                    if (base + k * 64u >= n) break;
                    const uint32_t tile_ix = first + i;
                    const bool valid = i < n && tile_ix < cfg.tiles_size;
                    int32_t v = valid ? cur[k] : 0;
                    // Nothing to add up where nothing was bumped: 64 zero backdrops behind a zero carry are their own prefix
                    // sums.  On a road map that is nearly every step (the two outlines of a stroke cancel within a tile or
                    // two, and path_count never writes the cancelled pairs), and the step is then its load alone -- no column
                    // arithmetic (an integer division), no six shuffle rounds.
                    if (__ballot(v != 0) == 0ull && carry == 0) continue;
                    const uint32_t col = i % width;
                    const int32_t own = v;
#pragma unroll
                    for (uint32_t d = 1; d < 64u; d <<= 1) {
                        const int32_t up = __shfl_up(v, (int)d);
                        if (lane >= d && col >= d) v += up;
                    }
                    if (col > lane) v += carry;  // the row began before this step's first lane
                    carry = __shfl(v, 63);
                    if (valid && v != own) tiles[tile_ix].backdrop = v;
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
    // grid sized for the pool capacity; workgroups beyond bump.lines exit after one load
    uint32_t grid = clamp_grid(f.cfg.lines_size, PATH_COUNT_CHUNK, 4096u * 4u / PATH_COUNT_LINES_PER_THREAD);
#if VK_PC_AGG
    hipLaunchKernelGGL(k_path_count_agg<PATH_COUNT_LINES_PER_THREAD>, dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.lines, f.paths, f.tiles, f.seg_counts);
    return;
#endif
    if (f.path_count_keep) hipLaunchKernelGGL(k_path_count<true>, dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.lines, f.paths, f.tiles, f.seg_counts);
    else hipLaunchKernelGGL(k_path_count<false>, dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.lines, f.paths, f.tiles, f.seg_counts);
}

void launch_backdrop(const Frame &f, hipStream_t s) {
    if (f.cfg.layout.n_paths == 0) return;
    // a workgroup per four paths (its waves share their row blocks), workgroups striding over the groups
    uint32_t n_wg = (f.cfg.layout.n_paths + 3u) / 4u;
    if (n_wg > 8192u) n_wg = 8192u;
    hipLaunchKernelGGL(k_backdrop, dim3(n_wg), dim3(256), 0, s, f.cfg, f.bump(), f.paths, f.tiles);
}

void launch_path_tiling(const Frame &f, hipStream_t s) {
    uint32_t grid = clamp_grid(f.cfg.seg_counts_size, 256u, 2048u);
    hipLaunchKernelGGL(k_path_tiling, dim3(grid), dim3(256), 0, s, f.cfg, f.bump(), f.seg_counts, f.lines, f.paths, f.tiles,
                       f.segments, f.ptcl);
}

}  // namespace vk
