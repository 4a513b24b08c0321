// SIMT emulator: a test-only stand-in for <hip/hip_runtime.h>.
//
// TEST INFRASTRUCTURE ONLY.  It lets `pytest -m "not gpu"` execute the *unmodified* kernel sources
// of vello_amd/csrc/engine/*.hip on the CPU (g++ -I tests/simt_emu), so indexing and algorithm
// errors are found in this GPU-less container instead of on a metered MI355X box.  It is never
// built into, loaded by, or a fallback for the product library libvello_hip.so: that library
// fails with VELLO_HIP_E_NO_DEVICE when there is no GPU.
//
// Model: workgroups run one after another on one OS thread; every work-item is a fiber with its
// own stack; __syncthreads() and the wave64 collectives (__shfl*, __ballot) are rendezvous points
// handled by a per-workgroup scheduler.  Wave collectives require all not-yet-exited lanes of the
// wave to arrive (the kernels only use them in wave-uniform control flow); a wave whose lanes
// wait at different kinds of rendezvous aborts with a diagnostic, which is how divergent-barrier
// bugs show up here.  Atomics are plain read-modify-writes.  Memory-model properties (fences,
// visibility across XCDs) are NOT modelled: those are verified on hardware.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define VELLO_SIMT_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 {
    unsigned x, y, z, w;
};
struct float4 {
    float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace simt_emu {
struct Idx {
    unsigned x, y, z;
};
extern Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void barrier();
// deposits `v` and returns after all live lanes of the wave arrived; vals[64] then holds every
// lane's value, *active the mask of participating lanes.  Returns this lane's index.
int wave_exchange(unsigned long long v, unsigned long long *vals, unsigned long long *active);
}  // namespace simt_emu

// what the kernels' wave_lds_sync() is here: every live lane of the wave arrives before any proceeds
#define VELLO_EMU_WAVE_RENDEZVOUS()                       \
    do {                                                  \
        unsigned long long v_[64], a_;                    \
        simt_emu::wave_exchange(0ull, v_, &a_);           \
    } while (0)

#define threadIdx simt_emu::g_threadIdx
#define blockIdx simt_emu::g_blockIdx
#define blockDim simt_emu::g_blockDim
#define gridDim simt_emu::g_gridDim

// ---------------- runtime API subset ----------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1 };
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated hip error"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)16 << 30; return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
    for (size_t y = 0; y < h; y++) std::memcpy((char *)d + y * dp, (const char *)s + y * sp, w);
    return hipSuccess;
}
enum { hipErrorPeerAccessAlreadyEnabled = 704, hipEventDisableTiming = 2 };
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) {
    return hipMemcpy2D(d, dp, s, sp, w, h, k);
}
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (void *)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    simt_emu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

// ---------------- device intrinsics subset ----------------
static inline void __syncthreads() { simt_emu::barrier(); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline long long clock64() { static long long t = 0; return t += 5; }  // (only the -DVELLO_*_PROF measurement builds read it)
static inline long long __double_as_longlong(double d) { long long u; std::memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; std::memcpy(&d, &u, 8); return d; }

template <class T>
static inline T __shfl(T v, int src) {
    unsigned long long bits = 0, vals[64], active;
    std::memcpy(&bits, &v, sizeof(T));
    simt_emu::wave_exchange(bits, vals, &active);
    T r;
    std::memcpy(&r, &vals[src & 63], sizeof(T));
    return r;
}
template <class T>
static inline T __shfl_up(T v, int delta) {
    unsigned long long bits = 0, vals[64], active;
    std::memcpy(&bits, &v, sizeof(T));
    int lane = simt_emu::wave_exchange(bits, vals, &active);
    if (lane < delta) return v;
    T r;
    std::memcpy(&r, &vals[lane - delta], sizeof(T));
    return r;
}
template <class T>
static inline T __shfl_down(T v, int delta) {
    unsigned long long bits = 0, vals[64], active;
    std::memcpy(&bits, &v, sizeof(T));
    int lane = simt_emu::wave_exchange(bits, vals, &active);
    if (lane + delta > 63) return v;
    T r;
    std::memcpy(&r, &vals[lane + delta], sizeof(T));
    return r;
}
template <class T>
static inline T __shfl_xor(T v, int mask) {
    unsigned long long bits = 0, vals[64], active;
    std::memcpy(&bits, &v, sizeof(T));
    int lane = simt_emu::wave_exchange(bits, vals, &active);
    T r;
    std::memcpy(&r, &vals[(lane ^ mask) & 63], sizeof(T));
    return r;
}
static inline int __builtin_amdgcn_readlane(int v, int lane) { return __shfl(v, lane); }
static inline int __builtin_amdgcn_readfirstlane(int v) {
    unsigned long long vals[64], active;
    simt_emu::wave_exchange((unsigned long long)(unsigned)v, vals, &active);
    return (int)(unsigned)vals[__builtin_ctzll(active)];
}
static inline unsigned long long __ballot(int pred) {
    unsigned long long vals[64], active;
    simt_emu::wave_exchange(pred ? 1ull : 0ull, vals, &active);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++)
        if (((active >> i) & 1ull) && vals[i]) m |= 1ull << i;
    return m;
}

template <class T>
static inline T atomicAdd(T *p, T v) { T o = *p; *p = (T)(o + v); return o; }
static inline unsigned atomicAdd(unsigned *p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
static inline int atomicAdd(int *p, unsigned v) { int o = *p; *p = (int)((unsigned)o + v); return o; }
template <class T>
static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T>
static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <class T>
static inline T atomicXor(T *p, T v) { T o = *p; *p = o ^ v; return o; }
template <class T>
static inline T atomicMin(T *p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <class T>
static inline T atomicMax(T *p, T v) { T o = *p; *p = std::max(o, v); return o; }

#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
template <typename T, typename V>
static inline T __hip_atomic_fetch_add(T *p, V v, int, int) { T o = *p; *p = (T)(o + (T)v); return o; }
