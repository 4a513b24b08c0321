// Microbenchmark: what a launch boundary costs against a grid barrier inside one launch (k_front, flatten.hip).
//   * a chain of N dependent launches of a kernel that does (next to) nothing, one stream: enqueue + wait, per launch;
//   * the same chain as a captured hipGraph, replayed;
//   * ONE launch of G workgroups that meets N grid barriers of k_front's kind (release-add, spin, acquire), G = 1 .. 256;
//   * a chain in which every kernel does one dependent global round trip per workgroup (what a small stage really is).
// Not part of the product.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_nothing(uint32_t *p) {
    if (p == nullptr) p[0] = 1u;  // (never)
}

// one dependent round trip: read a word the kernel before wrote, write the next
__global__ void __launch_bounds__(256) k_step(uint32_t *p, uint32_t i) {
    if (threadIdx.x == 0) p[(i + 1u) * 64u + blockIdx.x] = p[i * 64u + blockIdx.x] + 1u;
}

__global__ void __launch_bounds__(256) k_barriers(uint32_t *sync, uint32_t base, uint32_t n, uint32_t *p) {
    uint32_t target = base;
    for (uint32_t i = 0; i < n; i++) {
        if (threadIdx.x == 0) p[(i + 1u) * 64u + (blockIdx.x & 63u)] = p[i * 64u + (blockIdx.x & 63u)] + 1u;  // the stage: one round trip
        __syncthreads();
        if (gridDim.x != 1u) {
            target += gridDim.x;
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while ((int32_t)(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *buf, *sync;
    CK(hipMalloc(&buf, 64 * 64 * 4));
    CK(hipMalloc(&sync, 256));
    CK(hipMemset(buf, 0, 64 * 64 * 4));
    CK(hipMemset(sync, 0, 256));
    CK(hipDeviceSynchronize());
    const int REP = 300;
    for (int n : {1, 6, 12}) {
        for (int kind = 0; kind < 2; kind++) {
            for (int grid : {1, 16, 256}) {
                double best = 1e30, sum = 0;
                for (int r = 0; r < REP + 20; r++) {
                    const double t0 = now_us();
                    for (int i = 0; i < n; i++) {
                        if (kind == 0) hipLaunchKernelGGL(k_nothing, dim3(grid), dim3(256), 0, st, buf);
                        else hipLaunchKernelGGL(k_step, dim3(grid > 64 ? 64 : grid), dim3(256), 0, st, buf, (uint32_t)i);
                    }
                    CK(hipStreamSynchronize(st));
                    const double dt = now_us() - t0;
                    if (r >= 20) { sum += dt; if (dt < best) best = dt; }
                }
                std::printf("stream  %-8s chain of %2d launches x %3d workgroups: %7.1f us mean, %7.1f best -> %5.2f us per launch (mean)\n",
                            kind ? "step" : "nothing", n, grid, sum / REP, best, sum / REP / n);
            }
        }
    }
    // the chain as a graph
    for (int n : {6, 12}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_step, dim3(16), dim3(256), 0, st, buf, (uint32_t)i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double best = 1e30, sum = 0;
        for (int r = 0; r < REP + 20; r++) {
            const double t0 = now_us();
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            const double dt = now_us() - t0;
            if (r >= 20) { sum += dt; if (dt < best) best = dt; }
        }
        std::printf("graph   step     chain of %2d launches x  16 workgroups: %7.1f us mean, %7.1f best -> %5.2f us per launch (mean)\n", n, sum / REP, best, sum / REP / n);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // one launch, n barriers
    uint32_t base = 0;
    for (int n : {1, 6, 12}) {
        for (int grid : {1, 4, 16, 64, 256}) {
            double best = 1e30, sum = 0;
            for (int r = 0; r < REP + 20; r++) {
                const double t0 = now_us();
                hipLaunchKernelGGL(k_barriers, dim3(grid), dim3(256), 0, st, sync, base, (uint32_t)n, buf);
                CK(hipStreamSynchronize(st));
                const double dt = now_us() - t0;
                if (grid != 1) base += (uint32_t)(grid * n);
                if (r >= 20) { sum += dt; if (dt < best) best = dt; }
            }
            std::printf("barrier step     one launch of %3d workgroups, %2d stages: %7.1f us mean, %7.1f best -> %5.2f us per stage beyond the first launch\n", grid,
                        n, sum / REP, best, n > 1 ? (sum / REP) / n : 0.0);
        }
    }
    return 0;
}
