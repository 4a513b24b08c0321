// Does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let two kernels of ONE stream run side by side on gfx950?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this measures it.  Two one-workgroup kernels that each
// spin ~100 us: back to back they take ~200 us, side by side ~100.
//   hipcc --offload-arch=gfx950 -O2 scripts/calib/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned *out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out) out[blockIdx.x] = 1u;
}

static double run(hipStream_t s, unsigned flags, int n_kernels, unsigned long long ticks) {
    (void)hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, (unsigned *)nullptr);
    for (int i = 1; i < n_kernels; i++) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, ticks, (unsigned *)nullptr);
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int rate_khz = 0;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long ticks = (unsigned long long)rate_khz * 100ull / 1000ull;  // 100 us
    printf("wall clock %d kHz, %llu ticks per kernel\n", rate_khz, ticks);
    for (int rep = 0; rep < 3; rep++) {
        const double one = run(s, 0u, 1, ticks), in_order = run(s, 0u, 4, ticks), any_order = run(s, hipExtAnyOrderLaunch, 4, ticks);
        printf("1 kernel %.0f us | 4 kernels in order %.0f us | 4 kernels, 3 of them hipExtAnyOrderLaunch %.0f us\n", one, in_order, any_order);
    }
    return 0;
}
