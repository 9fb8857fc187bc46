// calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the observation kernel
// (MI355X_MICROARCH.md, section HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern before trusting an absolute").  Every kernel touches a KNOWN number of bytes / cache lines of a buffer far larger than
// the L2 and the Infinity Cache:
//   k_stream_read   16 B per lane, coalesced           (the guide's reference pattern: FETCH_SIZE reports 1/2 of the bytes)
//   k_gather_read   one 4-byte word per lane, every access in a line of its own (64-byte stride, shuffled): the dm / seg /
//                   hop8 / item gathers of the observation kernel
//   k_gather16_read 16 contiguous 4-byte words per lane at a shuffled 64-byte aligned place: a chunk of a key's item list
//   k_stream_write  16 B per lane, coalesced           (the observation tensors)
//   k_scatter_write one 4-byte word per lane, every store in a line of its own: the prediction items of large maps
// Build: hipcc --offload-arch=gfx950 -O3 calib.hip -o calib ; run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }

__global__ void k_stream_read(const uint4 *src, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// n_lines lines of 64 B; access a visits line perm(a) (a bijection on [0, n_lines) for a power of two: odd multiplier + xor)
__device__ __forceinline__ size_t perm(size_t a, size_t mask) { return ((a * 0x9E3779B1ull) ^ (a >> 7)) & mask; }
__global__ void k_gather_read(const uint32_t *src, size_t n_acc, size_t line_mask, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < n_acc; a += (size_t)gridDim.x * blockDim.x) acc ^= src[perm(a, line_mask) * 16 + (a & 15)];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_gather16_read(const uint32_t *src, size_t n_acc, size_t line_mask, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < n_acc; a += (size_t)gridDim.x * blockDim.x) {
        const uint32_t *p = src + perm(a, line_mask) * 16;
#pragma unroll
        for (int q = 0; q < 16; q++) acc ^= p[q];
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream_write(uint4 *dst, size_t n16) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) dst[k] = make_uint4((uint32_t)k, 1u, 2u, 3u);
}
__global__ void k_scatter_write(uint32_t *dst, size_t n_acc, size_t line_mask) {
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < n_acc; a += (size_t)gridDim.x * blockDim.x) dst[perm(a, line_mask) * 16 + (a & 15)] = (uint32_t)a;
}

int main() {
    const size_t bytes = (size_t)4 << 30;           // 4 GiB: far beyond L2 (8 x 4 MiB) and the Infinity Cache (256 MiB)
    const size_t n_lines = bytes / 64, line_mask = n_lines - 1;
    const size_t n_acc = (size_t)1 << 24;           // 16 Mi accesses, each in a line of its own (n_lines = 64 Mi)
    void *buf; uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    const size_t stream_bytes = (size_t)1 << 30;    // 1 GiB streamed
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_stream_read, grid, block, 0, 0, (const uint4 *)buf, stream_bytes / 16, sink);
        hipLaunchKernelGGL(k_gather_read, grid, block, 0, 0, (const uint32_t *)buf, n_acc, line_mask, sink);
        hipLaunchKernelGGL(k_gather16_read, grid, block, 0, 0, (const uint32_t *)buf, n_acc, line_mask, sink);
        hipLaunchKernelGGL(k_stream_write, grid, block, 0, 0, (uint4 *)buf, stream_bytes / 16);
        hipLaunchKernelGGL(k_scatter_write, grid, block, 0, 0, (uint32_t *)buf, n_acc, line_mask);
        CHECK(hipDeviceSynchronize());
    }
    printf("expected per launch: stream_read %zu B; gather_read %zu accesses (x 4 B used, x 64 B lines = %zu B); gather16_read %zu lines x 64 B = %zu B; "
           "stream_write %zu B; scatter_write %zu accesses (x 4 B, x 64 B lines = %zu B)\n", stream_bytes, n_acc, n_acc * 64, n_acc, n_acc * 64, stream_bytes, n_acc, n_acc * 64);
    return 0;
}
