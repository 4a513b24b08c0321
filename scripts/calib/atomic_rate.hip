// Microbenchmark: aggregate rate of device-scope global atomics on MI355X, by footprint, locality and whether the old
// value is returned.  Informs the bound of k_path_count / k_flatten.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode 0: every lane its own random word; 1: the 64 lanes of a wave hit 64 words of 4 random 64-byte lines-of-16;
// 2: lanes of a wave hit consecutive words (one 256-byte run at a random place)
template <bool RET, int MODE>
__global__ void k_atomics(uint32_t *buf, uint32_t words_mask, uint32_t per_thread, uint32_t *sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        uint32_t ix;
        if (MODE == 0) ix = hash32(gid * per_thread + i);
        else if (MODE == 1) ix = hash32((gid >> 4) * per_thread + i) * 16u + (gid & 15u);
        else ix = hash32((gid >> 6) * per_thread + i) * 64u + (gid & 63u);
        ix &= words_mask;
        if (RET) acc += atomicAdd(&buf[ix], 1u);
        else atomicAdd(&buf[ix], 1u);
    }
    if (RET && acc == 0xffffffffu) *sink = acc;
}

template <bool RET, int MODE>
static void run(const char *name, uint32_t *buf, uint32_t words, uint32_t *sink) {
    const uint32_t threads = 1u << 20, per_thread = 16u;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_atomics<RET, MODE>), dim3(threads / 256), dim3(256), 0, 0, buf, words - 1u, per_thread, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-28s footprint %7.1f MB: %7.2f G atomics/s\n", name, words * 4.0 / 1e6, threads * (double)per_thread / (best * 1e-3) / 1e9);
}

int main() {
    uint32_t *buf, *sink;
    const uint32_t max_words = 1u << 25;  // 128 MB
    hipMalloc((void **)&buf, max_words * 4ull); hipMemset(buf, 0, max_words * 4ull);
    hipMalloc((void **)&sink, 4);
    for (uint32_t words : {1u << 18, 1u << 21, 1u << 24, 1u << 25}) {
        run<false, 0>("scattered, no return", buf, words, sink);
        run<true, 0>("scattered, returning", buf, words, sink);
        run<false, 1>("16-word lines, no return", buf, words, sink);
        run<true, 1>("16-word lines, returning", buf, words, sink);
        run<false, 2>("64 consecutive, no return", buf, words, sink);
        run<true, 2>("64 consecutive, returning", buf, words, sink);
    }
    return 0;
}
