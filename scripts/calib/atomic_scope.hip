// Microbenchmark + correctness probe: global atomics that stay inside ONE XCD's L2.
//
// MI355X has eight XCDs with an L2 each.  A device-scope (agent) atomic is carried out at the memory side so that all XCDs
// agree on it: 2.7e10 requests per second chip-wide (atomic_rate.hip), the bound of k_path_count.  An atomic of WORKGROUP
// scope is carried out in the issuing XCD's L2.  If every address is only ever touched from one XCD during a launch, that is
// enough -- the L2 is written back at the end of the kernel -- and the question is what it buys and whether it is safe:
//   * which XCD does workgroup b run on (HW_REG_XCC_ID against blockIdx.x)?
//   * rate of workgroup-scope atomics, addresses partitioned by the XCD that issues them, against agent scope;
//   * are the results right: final counts add up, the returned values of an address are a permutation of 0 .. n-1;
//   * and unpartitioned (every XCD on every address) for contrast: expected to lose updates.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

__global__ void k_where(uint32_t *xcc_of_block) {
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = xcc_id();
}

// SCOPE 0: agent, 1: workgroup.  PART: addresses from the region of the issuing XCD (words / 8 each).
// MODE 0: every lane its own random word; 1: a wave hits 4 random 16-word lines (the tile atomics of path_count look like this)
template <int SCOPE, bool PART, bool RET, int MODE>
__global__ void __launch_bounds__(256) k_atomics(uint32_t *buf, uint32_t region_words, uint32_t per_thread, unsigned long long *ret_sum) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t region = PART ? (xcc_id() & 7u) : (hash32(gid ^ 0x9e3779b9u) & 7u);
    unsigned long long acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        uint32_t ix;
        if (MODE == 0) ix = hash32(gid * per_thread + i);
        else ix = hash32((gid >> 4) * per_thread + i) * 16u + (gid & 15u);
        ix = (ix & (region_words - 1u)) + region * region_words;
        if (RET) {
            if (SCOPE == 0) acc += __hip_atomic_fetch_add(&buf[ix], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else acc += __hip_atomic_fetch_add(&buf[ix], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (SCOPE == 0) (void)__hip_atomic_fetch_add(&buf[ix], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else (void)__hip_atomic_fetch_add(&buf[ix], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (RET && ret_sum) {
        // (verification launches only: one device-scope atomic per wave)
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if ((threadIdx.x & 63u) == 0u) atomicAdd(ret_sum, acc);
    }
}

template <int SCOPE, bool PART, bool RET, int MODE>
static void run(const char *name, uint32_t *buf, uint32_t words, unsigned long long *d_ret) {
    const uint32_t threads = 1u << 20, per_thread = 16u;
    const uint32_t region_words = words / 8u;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_atomics<SCOPE, PART, RET, MODE>), dim3(threads / 256), dim3(256), 0, 0, buf, region_words, per_thread, nullptr);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    // verification launch on a cleared buffer
    hipMemset(buf, 0, words * 4ull);
    hipMemset(d_ret, 0, 8);
    hipLaunchKernelGGL((k_atomics<SCOPE, PART, RET, MODE>), dim3(threads / 256), dim3(256), 0, 0, buf, region_words, per_thread, RET ? d_ret : nullptr);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(words);
    hipMemcpy(h.data(), buf, words * 4ull, hipMemcpyDeviceToHost);
    unsigned long long total = 0, want_ret = 0, got_ret = 0;
    for (uint32_t v : h) { total += v; want_ret += (unsigned long long)v * (v - 1u) / 2u; }
    hipMemcpy(&got_ret, d_ret, 8, hipMemcpyDeviceToHost);
    const unsigned long long ops = (unsigned long long)threads * per_thread;
    const bool ok = total == ops && (!RET || got_ret == want_ret);
    printf("%-46s %6.1f MB: %7.2f G atomics/s   counts %s (%llu of %llu)%s\n", name, words * 4.0 / 1e6,
           threads * (double)per_thread / (best * 1e-3) / 1e9, total == ops ? "add up" : "LOST", total, ops,
           RET ? (got_ret == want_ret ? ", returns a permutation" : ", returns WRONG") : "");
    (void)ok;
}

int main() {
    uint32_t *buf, *d_where;
    unsigned long long *d_ret;
    const uint32_t max_words = 1u << 25;  // 128 MB
    hipMalloc((void **)&buf, max_words * 4ull); hipMemset(buf, 0, max_words * 4ull);
    hipMalloc((void **)&d_ret, 8);
    const uint32_t nb = 4096;
    hipMalloc((void **)&d_where, nb * 4);
    for (int trial = 0; trial < 3; trial++) {
        hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, 0, d_where);
        hipDeviceSynchronize();
        std::vector<uint32_t> w(nb);
        hipMemcpy(w.data(), d_where, nb * 4, hipMemcpyDeviceToHost);
        uint32_t same = 0, hist[16] = {};
        for (uint32_t b = 0; b < nb; b++) { same += (w[b] == (b & 7u)); hist[w[b] & 15u]++; }
        printf("trial %d: blocks with XCC_ID == blockIdx %% 8: %u of %u; per XCC:", trial, same, nb);
        for (int x = 0; x < 16; x++) if (hist[x]) printf(" %d:%u", x, hist[x]);
        printf("; first 16:");
        for (int b = 0; b < 16; b++) printf(" %u", w[b]);
        printf("\n");
    }
    for (uint32_t words : {1u << 21, 1u << 24, 1u << 25}) {
        run<0, true, false, 0>("agent, by XCD, scattered, no return", buf, words, d_ret);
        run<0, true, true, 0>("agent, by XCD, scattered, returning", buf, words, d_ret);
        run<1, true, false, 0>("workgroup scope, by XCD, scattered, no return", buf, words, d_ret);
        run<1, true, true, 0>("workgroup scope, by XCD, scattered, returning", buf, words, d_ret);
        run<0, true, true, 1>("agent, by XCD, 16-word lines, returning", buf, words, d_ret);
        run<1, true, false, 1>("workgroup scope, by XCD, 16-word lines, no ret", buf, words, d_ret);
        run<1, true, true, 1>("workgroup scope, by XCD, 16-word lines, ret", buf, words, d_ret);
        run<1, false, true, 0>("workgroup scope, ANY XCD, scattered, returning", buf, words, d_ret);
    }
    return 0;
}
