// Single-pass decoupled look-back over NF-field u32 monoids (Merrill & Garland), shaped for gfx950:
// one wave64 inspects 64 predecessor partitions per step with 8-byte granule loads, ballot picks
// the nearest published prefix, a butterfly reduction sums the aggregates in front of it.
// Replaces the reference's reduce / reduce2 / scan1 / scan dispatch chain
// (vello/src/render.rs:250-294, vello_shaders/shader/pathtag_*.wgsl, draw_reduce.wgsl); the
// monoids are integer sums, so the result is bit-identical whatever the scan order.
#pragma once
#include "engine.h"

namespace vk {

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long *p, uint32_t status, uint32_t value) {
    __hip_atomic_store(p, ((unsigned long long)status << 32) | (unsigned long long)value, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// Must be called by all 64 lanes of ONE wave (convergent).  `agg` is the partition aggregate
// (same value in every lane).  On return `excl` holds the exclusive prefix of partition `part`
// in every lane and the inclusive prefix has been published.
template <int NF>
__device__ __forceinline__ void decoupled_lookback(unsigned long long *state, uint32_t part, const uint32_t (&agg)[NF],
                                                   uint32_t (&excl)[NF], uint32_t *failed_flag) {
    const int lane = threadIdx.x & 63;
    unsigned long long *my_agg = state + (size_t)part * 2u * NF;
    unsigned long long *my_pre = my_agg + NF;
#pragma unroll
    for (int f = 0; f < NF; f++) excl[f] = 0u;
    if (part != 0u) {
        // publish the aggregate so that successors can skip over this partition
#pragma unroll
        for (int f = 0; f < NF; f++)
            if (lane == f) granule_store(my_agg + f, SCAN_STATUS_AGG, agg[f]);
        int base = (int)part - 1;
        uint32_t spins = 0;
        while (base >= 0) {
            int q = base - lane;
            uint32_t st = SCAN_STATUS_PREFIX;
            uint32_t v[NF];
#pragma unroll
            for (int f = 0; f < NF; f++) v[f] = 0u;
            if (q >= 0) {
                const unsigned long long *qa = state + (size_t)q * 2u * NF;
                const unsigned long long *qp = qa + NF;
                bool ok = true;
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    unsigned long long g = granule_load(qp + f);
                    ok = ok && (uint32_t)(g >> 32) == SCAN_STATUS_PREFIX;
                    v[f] = (uint32_t)g;
                }
                if (!ok) {
                    ok = true;
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        unsigned long long g = granule_load(qa + f);
                        ok = ok && (uint32_t)(g >> 32) == SCAN_STATUS_AGG;
                        v[f] = (uint32_t)g;
                    }
                    st = ok ? SCAN_STATUS_AGG : 0u;
                }
            }
            unsigned long long b_prefix = __ballot(st == SCAN_STATUS_PREFIX);
            unsigned long long b_invalid = __ballot(st == 0u);
            int first_prefix = b_prefix ? (__ffsll((long long)b_prefix) - 1) : 64;
            unsigned long long need = first_prefix >= 63 ? ~0ull : ((1ull << (first_prefix + 1)) - 1ull);
            if (b_invalid & need) {
                if (++spins > SPIN_LIMIT) {
                    if (lane == 0) atomicOr(failed_flag, FAILED_INTERNAL);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            bool take = ((need >> lane) & 1ull) != 0ull;
#pragma unroll
            for (int f = 0; f < NF; f++) excl[f] += wave_sum_u32(take ? v[f] : 0u);
            if (first_prefix < 64) break;
            base -= 64;
        }
    }
#pragma unroll
    for (int f = 0; f < NF; f++)
        if (lane == f) granule_store(my_pre + f, SCAN_STATUS_PREFIX, excl[f] + agg[f]);
}

}  // namespace vk
