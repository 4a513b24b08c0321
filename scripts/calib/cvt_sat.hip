// Checks on the device that v_cvt_i32_f32 / v_cvt_u32_f32 are WGSL's saturating conversions (common.h: f2i / f2u use the
// bare instruction): every f32 bit pattern of a strided sweep plus the edge values, against the guarded C form.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

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
__global__ void k_check(uint32_t stride, unsigned long long *bad, uint32_t *first_bad) {
    const unsigned long long n = (0x100000000ull + stride - 1) / stride;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)(i * stride);
        const float f = __uint_as_float(bits);
        uint32_t u;
        int32_t s;
        asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(f));
        asm("v_cvt_i32_f32 %0, %1" : "=v"(s) : "v"(f));
        if (u != ref_f2u(f) || s != ref_f2i(f)) {
            if (atomicAdd(bad, 1ull) == 0ull) *first_bad = bits;
        }
    }
}
int main() {
    unsigned long long *bad, hb = 0;
    uint32_t *first, hf = 0;
    hipMalloc((void **)&bad, 8);
    hipMalloc((void **)&first, 4);
    hipMemset(bad, 0, 8);
    hipMemset(first, 0, 4);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, 1u, bad, first);  // all 2^32 bit patterns
    hipDeviceSynchronize();
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("v_cvt_{u32,i32}_f32 vs the guarded conversions over all 2^32 f32 bit patterns: %llu mismatches", hb);
    if (hb) printf(" (first at bits 0x%08x)", hf);
    printf("\n");
    return hb != 0;
}
