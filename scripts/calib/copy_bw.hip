// Measurement helper for bench.py (NOT part of libvello_hip.so): the float4 device-to-device copy MI355X_MICROARCH.md quotes
// for "achievable HBM bandwidth" (6.29 TB/s), timed with HIP events on this GPU in this run.
//   extern "C" double copy_bw_gbps(int device, size_t bytes, int reps)  ->  (bytes read + bytes written) / best time, GB/s;
//   negative = a HIP error code
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

// four 16-byte loads in flight per thread and iteration, a workgroup on 16 consecutive KB
__global__ void __launch_bounds__(256) k_copy16(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    const size_t per_block = 256u * 4u;
    for (size_t base = (size_t)blockIdx.x * per_block; base < n; base += (size_t)gridDim.x * per_block) {
        const size_t i = base + threadIdx.x;
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + 256u * k < n ? src[i + 256u * k] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + 256u * k < n) dst[i + 256u * k] = v[k];
    }
}

extern "C" double copy_bw_gbps(int device, size_t bytes, int reps) {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return -(double)e;
    float4 *a = nullptr, *b = nullptr;
    if ((e = hipMalloc((void **)&a, bytes)) != hipSuccess) return -(double)e;
    if ((e = hipMalloc((void **)&b, bytes)) != hipSuccess) {
        (void)hipFree(a);
        return -(double)e;
    }
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t n = bytes / sizeof(float4);
    double best = 0.0;
    // (the best of a few grid shapes: workgroups per CU resident at once vs one pass over the buffer)
    const unsigned grids[4] = {256u * 8u, 256u * 16u, 256u * 32u, (unsigned)((n + 1023u) / 1024u)};
    for (unsigned grid : grids) {
        for (int r = 0; r < reps + 2; r++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, a, b, n);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2 && ms > 0.f) {
                const double g = 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
                if (g > best) best = g;
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return best;
}
