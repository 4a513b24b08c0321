// Microbenchmark (round 6): the rate of returning device-scope atomics on ONE word from many workgroups -- what k_path_count's
// chunk reservation (atomicAdd(&bump->seg_counts, total), one per chunk of 1 024 lines) and flatten's line reservations are made of.
// grid workgroups of 256 threads, lane 0 of each adds `per_wg` times: (a) each add waits for the answer of the one before (a chain,
// as a workgroup that needs the answer before it goes on), (b) adds spaced by `gap` ns of arithmetic (a workgroup that computes
// between two reservations).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void k_same(uint32_t *word, uint32_t per_wg, uint32_t gap_iters, uint32_t *sink) {
    if (threadIdx.x != 0) return;
    uint32_t acc = 0;
    float f = (float)blockIdx.x;
    for (uint32_t i = 0; i < per_wg; i++) {
        acc += atomicAdd(word, 1u + (acc & 1u));  // (the next add depends on this one's answer)
        for (uint32_t g = 0; g < gap_iters; g++) f = f * 1.0001f + 0.5f;
    }
    if (acc == 0xffffffffu || f == 12345.0f) *sink = acc;
}
// the same number of adds spread over `n_words` words 256 bytes apart (one per workgroup modulo n_words)
__global__ void k_spread(uint32_t *words, uint32_t n_words, uint32_t per_wg, uint32_t *sink) {
    if (threadIdx.x != 0) return;
    uint32_t acc = 0;
    uint32_t *w = words + (blockIdx.x % n_words) * 64u;
    for (uint32_t i = 0; i < per_wg; i++) acc += atomicAdd(w, 1u + (acc & 1u));
    if (acc == 0xffffffffu) *sink = acc;
}

int main() {
    uint32_t *buf, *sink;
    hipMalloc((void **)&buf, 1u << 20); hipMemset(buf, 0, 1u << 20);
    hipMalloc((void **)&sink, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto time_it = [&](auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        return best;
    };
    const uint32_t per_wg = 64u;
    for (uint32_t grid : {1u, 8u, 64u, 256u, 768u, 3072u}) {
        float ms = time_it([&] { hipLaunchKernelGGL(k_same, dim3(grid), dim3(256), 0, 0, buf, per_wg, 0u, sink); });
        printf("one word, %4u workgroups x %u chained returning adds: %8.1f us, %6.1f ns per add (chip-wide), %6.2f us per add of a workgroup\n", grid, per_wg,
               ms * 1e3, ms * 1e6 / (grid * per_wg), ms * 1e3 / per_wg);
    }
    for (uint32_t gap : {1000u, 4000u, 16000u}) {
        float ms = time_it([&] { hipLaunchKernelGGL(k_same, dim3(768), dim3(256), 0, 0, buf, 16u, gap, sink); });
        float ms0 = time_it([&] { hipLaunchKernelGGL(k_same, dim3(1), dim3(256), 0, 0, buf, 16u, gap, sink); });
        printf("one word, 768 workgroups x 16 adds with %5u fma between: %8.1f us (one workgroup alone: %8.1f us), %6.1f ns per add chip-wide\n", gap, ms * 1e3, ms0 * 1e3,
               ms * 1e6 / (768 * 16));
    }
    for (uint32_t n_words : {1u, 8u, 64u, 768u}) {
        float ms = time_it([&] { hipLaunchKernelGGL(k_spread, dim3(768), dim3(256), 0, 0, buf, n_words, per_wg, sink); });
        printf("%4u words, 768 workgroups x %u chained returning adds: %8.1f us, %6.1f ns per add chip-wide\n", n_words, per_wg, ms * 1e3, ms * 1e6 / (768 * per_wg));
    }
    return 0;
}
