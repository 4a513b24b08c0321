// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns this engine uses
// (MI355X_MICROARCH.md "HBM": only the wide streaming read is calibrated there).  Every kernel moves a KNOWN
// number of bytes; scripts/pmc_summary.py divides the counter by it.  Buffers are 512 MiB (> 256 MiB L3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t BYTES = 512ull << 20;

// 16 B per lane, fully coalesced
__global__ void calib_read16(const uint4 *__restrict__ src, uint32_t *sink, size_t n) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 4 B per lane, coalesced
__global__ void calib_read4(const uint32_t *__restrict__ src, uint32_t *sink, size_t n) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_write16(uint4 *dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// fine's target pattern: one wave = one 16x16 RGBA8 tile, lane -> (row = lane/4, 16-byte quarter = lane%4),
// i.e. sixteen 64-byte row pieces `stride` bytes apart per store instruction
__global__ void calib_write_tile(uint8_t *dst, uint32_t width_tiles, uint32_t height_tiles, uint32_t stride) {
    for (uint32_t t = blockIdx.x; t < width_tiles * height_tiles; t += gridDim.x) {
        uint32_t tx = t % width_tiles, ty = t / width_tiles;
        uint32_t row = threadIdx.x >> 2, q = threadIdx.x & 3u;
        uint8_t *p = dst + (size_t)(ty * 16u + row) * stride + (size_t)tx * 64u + q * 16u;
        *reinterpret_cast<uint4 *>(p) = make_uint4(t, 1, 2, 3);
    }
}
// 24-byte records (LineSoup / Segment), one record per lane, written as 3 x 8 B
__global__ void calib_write24(uint2 *dst, size_t n_rec) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_rec; i += (size_t)gridDim.x * blockDim.x) {
        dst[3 * i] = make_uint2((uint32_t)i, 0);
        dst[3 * i + 1] = make_uint2(1, 2);
        dst[3 * i + 2] = make_uint2(3, 4);
    }
}
// 24-byte records gathered at a pseudo-random index (path_tiling's line gather)
__global__ void calib_gather24(const uint2 *__restrict__ src, uint32_t *sink, size_t n_rec, size_t n_gather) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_gather; i += (size_t)gridDim.x * blockDim.x) {
        size_t j = (i * 2654435761ull) % n_rec;
        uint2 a = src[3 * j], b = src[3 * j + 1], c = src[3 * j + 2];
        acc ^= a.x ^ b.x ^ c.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&b, BYTES) != hipSuccess) return 1;
    hipMemset(a, 1, BYTES);
    hipMemset(b, 0, BYTES);
    uint32_t *sink = (uint32_t *)b;
    dim3 g(256 * 16), blk(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read16, g, blk, 0, 0, (const uint4 *)a, sink, BYTES / 16);
        hipLaunchKernelGGL(calib_read4, g, blk, 0, 0, (const uint32_t *)a, sink, BYTES / 4);
        hipLaunchKernelGGL(calib_write16, g, blk, 0, 0, (uint4 *)b, BYTES / 16);
        // 8192 x 16384 px RGBA8 = 512 MiB
        hipLaunchKernelGGL(calib_write_tile, dim3(256 * 32), dim3(64), 0, 0, (uint8_t *)b, 512u, 1024u, 8192u * 4u);
        hipLaunchKernelGGL(calib_write24, g, blk, 0, 0, (uint2 *)b, BYTES / 24);
        hipLaunchKernelGGL(calib_gather24, g, blk, 0, 0, (const uint2 *)a, sink, BYTES / 24, BYTES / 24);
    }
    hipDeviceSynchronize();
    printf("bytes_per_kernel %zu (write24/gather24: %zu)\n", BYTES, (BYTES / 24) * 24);
    return 0;
}
