// TEST-ONLY device check: common.h gives f2u / f2i, row_shr / row_shr0, bcast_byte3, lane_value, wave_shfl,
// wave_read, mask_rank_below and the two wave scans one body for the GPU (inline asm, DPP, v_perm, v_readlane, ds_bpermute,
// v_mbcnt) and another for the SIMT emulator (portable C).  The emulator parity suite therefore never runs what ships; this
// library runs the GPU bodies on the device against the portable forms, so that `pytest -m gpu` catches a toolchain or ISA
// difference.  Built by __graft_entry__.build() into tests/device_checks/libvello_devcheck.so; never loaded by vello_amd.
#include "../../vello_amd/csrc/engine/common.h"
#include <cstdio>

namespace {

__device__ uint32_t ref_f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
__device__ int32_t ref_f2i(float f) {
    if (f != f) return 0;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    if (f >= 2147483648.0f) return 0x7fffffff;
    return (int32_t)f;
}

// every f32 bit pattern i * stride (+ the edge values handed in `edges`): common.h's f2u / f2i against the guarded conversions
__global__ void k_cvt(uint32_t stride, const uint32_t *edges, uint32_t n_edges, unsigned long long *bad) {
    const unsigned long long n = (0x100000000ull + stride - 1) / stride + n_edges;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t bits = i < n_edges ? edges[i] : (uint32_t)((i - n_edges) * stride);
        const float f = __uint_as_float(bits);
        if (vk::f2u(f) != ref_f2u(f)) atomicAdd(&bad[0], 1ull);
        if (vk::f2i(f) != ref_f2i(f)) atomicAdd(&bad[1], 1ull);
    }
}

// one wave64 per block; values through LDS give the portable answers
__global__ void __launch_bounds__(64) k_wave(uint32_t seed, unsigned long long *bad) {
    __shared__ uint32_t sh[64];
    const uint32_t lane = threadIdx.x;
    // a cheap hash for per-lane test values, different per block
    uint32_t v = (lane + 1u) * 0x9e3779b1u ^ (blockIdx.x + seed) * 0x85ebca6bu;
    v ^= v >> 15; v *= 0x2c1b3c6du; v ^= v >> 12;
    if ((blockIdx.x & 3u) == 1u) v &= 0xffu;        // small values (the scans' sums do not wrap)
    if ((blockIdx.x & 3u) == 2u) v = lane;           // a recognisable pattern
    sh[lane] = v;
    __syncthreads();
    const uint32_t row_pos = lane & 15u;
    auto at = [&](uint32_t l) { return sh[l & 63u]; };
    // Every cross-lane primitive is evaluated with ALL 64 lanes active, as the kernels use them (a DPP move or a ds_bpermute
    // whose source lane is switched off returns the old / a zero value: under `if (row_pos >= K && ...)` the lane at position K
    // would read a disabled lane); only the comparisons are per lane.
    const uint32_t s1 = vk::row_shr<1>(v), s2 = vk::row_shr<2>(v), s3 = vk::row_shr<3>(v);
    const uint32_t z1 = vk::row_shr0<1>(v), z2 = vk::row_shr0<2>(v), z3 = vk::row_shr0<3>(v);
    const uint32_t lv0 = vk::lane_value<0>(v), lv12 = vk::lane_value<12>(v), lv63 = vk::lane_value<63>(v);
    const uint32_t src = (v >> 7) & 63u;
    const uint32_t shf = vk::wave_shfl(v, src);
    const uint32_t usrc = (at(5u) >> 3) & 63u;  // a wave-uniform lane number
    const uint32_t rd = vk::wave_read(v, usrc);
    const unsigned long long m = __ballot((v & 1u) != 0u);
    const uint32_t rank = vk::mask_rank_below(m, lane), scan_add = vk::wave_incl_scan_u32(v, (int)lane), scan_max = vk::wave_incl_scan_max_u32(v, (int)lane);
    // row_shr<K> (defined for lanes whose source is in the row), row_shr0<K> (0 elsewhere)
    if (row_pos >= 1u && s1 != at(lane - 1u)) atomicAdd(&bad[2], 1ull);
    if (row_pos >= 2u && s2 != at(lane - 2u)) atomicAdd(&bad[2], 1ull);
    if (row_pos >= 3u && s3 != at(lane - 3u)) atomicAdd(&bad[2], 1ull);
    if (z1 != (row_pos >= 1u ? at(lane - 1u) : 0u)) atomicAdd(&bad[3], 1ull);
    if (z2 != (row_pos >= 2u ? at(lane - 2u) : 0u)) atomicAdd(&bad[3], 1ull);
    if (z3 != (row_pos >= 3u ? at(lane - 3u) : 0u)) atomicAdd(&bad[3], 1ull);
    if (vk::bcast_byte3(v) != (v >> 24) * 0x1010101u) atomicAdd(&bad[4], 1ull);
    if (lv0 != at(0u) || lv12 != at(12u) || lv63 != at(63u)) atomicAdd(&bad[5], 1ull);
    if (shf != at(src)) atomicAdd(&bad[6], 1ull);
    if (rd != at(usrc)) atomicAdd(&bad[7], 1ull);
    uint32_t below = 0u;
    for (uint32_t l = 0; l < lane; l++) below += at(l) & 1u;
    if (rank != below) atomicAdd(&bad[8], 1ull);
    uint32_t sum = 0u, mx = 0u;
    for (uint32_t l = 0; l <= lane; l++) {
        sum += at(l);
        mx = at(l) > mx ? at(l) : mx;
    }
    if (scan_add != sum) atomicAdd(&bad[9], 1ull);
    if (scan_max != mx) atomicAdd(&bad[10], 1ull);
}

}  // namespace

// bad[0..10]: mismatches of f2u, f2i, row_shr, row_shr0, bcast_byte3, lane_value, wave_shfl, wave_read, mask_rank_below,
// wave_incl_scan_u32, wave_incl_scan_max_u32.  Returns 0 when the checks ran (whatever they found), a hipError_t otherwise.
extern "C" int vello_devcheck_primitives(uint32_t cvt_stride, unsigned long long *bad_out) {
    static const uint32_t edges[] = {
        0x00000000u, 0x80000000u, 0x3f800000u, 0xbf800000u, 0x3f7fffffu, 0x00000001u, 0x80000001u,   // 0, -0, 1, -1, 1-ulp, denormals
        0x4effffffu, 0x4f000000u, 0x4f000001u, 0xcf000000u, 0xcf000001u, 0xceffffffu,                 // around 2^31 and -2^31
        0x4f7fffffu, 0x4f800000u, 0x4f800001u, 0xcf800000u,                                           // around 2^32
        0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00000u, 0x7f800001u, 0xffa00000u};  // max, inf, NaNs
    const uint32_t n_edges = sizeof(edges) / sizeof(edges[0]);
    unsigned long long *bad = nullptr;
    uint32_t *d_edges = nullptr;
    hipError_t e = hipMalloc((void **)&bad, 11 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    e = hipMalloc((void **)&d_edges, sizeof(edges));
    if (e != hipSuccess) return (int)e;
    (void)hipMemset(bad, 0, 11 * sizeof(unsigned long long));
    (void)hipMemcpy(d_edges, edges, sizeof(edges), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(4096), dim3(256), 0, 0, cvt_stride ? cvt_stride : 1u, d_edges, n_edges, bad);
    hipLaunchKernelGGL(k_wave, dim3(1024), dim3(64), 0, 0, 12345u, bad);
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(bad_out, bad, 11 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(bad);
    (void)hipFree(d_edges);
    return (int)e;
}
