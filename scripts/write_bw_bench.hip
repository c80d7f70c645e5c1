// How fast can MI355X WRITE to HBM, and how does the rate depend on the number of concurrent streams?  (The transposed keys and the
// class codes are 805 MB of writes per step; k_bin_cols and k_transpose_count both look write-bound at ~2.3-2.8 TB/s.)
//   pattern A: one contiguous grid-stride fill (every wave writes 1 KiB, consecutive waves consecutive KiB)
//   pattern B: S independent streams (one per block), each written sequentially by its block in 1 KiB..8 KiB pieces
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/write_bw_bench.hip -o scripts/bin/write_bw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(1024) void fill_a(uint4 *p, size_t n16) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(1024) void fill_b(uint4 *p, size_t n16_per_stream) {
    uint4 *q = p + static_cast<size_t>(blockIdx.x) * n16_per_stream;
    for (size_t i = threadIdx.x; i < n16_per_stream; i += blockDim.x) q[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(1024) void copy_a(const uint4 *s, uint4 *p, size_t n16) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = s[i];
}
int main() {
    const size_t bytes = size_t(1) << 30;
    uint4 *p, *s; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&s, bytes)); CK(hipMemset(s, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    for (int blocks : {256, 512, 1024, 2048, 8192}) {
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL(fill_a, dim3(blocks), dim3(1024), 0, 0, p, bytes / 16); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); }
        CK(hipEventElapsedTime(&ms, a, b));
        printf("fill, one contiguous stream, %5d blocks: %.1f us  %.2f TB/s\n", blocks, ms * 1e3, bytes / ms / 1e9);
    }
    for (int streams : {256, 512, 1024, 4096}) {
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL(fill_b, dim3(streams), dim3(1024), 0, 0, p, bytes / 16 / streams); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); }
        CK(hipEventElapsedTime(&ms, a, b));
        printf("fill, %5d block-private streams:        %.1f us  %.2f TB/s\n", streams, ms * 1e3, bytes / ms / 1e9);
    }
    for (int blocks : {512, 2048}) {
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL(copy_a, dim3(blocks), dim3(1024), 0, 0, s, p, bytes / 16); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); }
        CK(hipEventElapsedTime(&ms, a, b));
        printf("copy (1 GiB read + 1 GiB write), %5d blocks: %.1f us  %.2f TB/s read+write\n", blocks, ms * 1e3, 2.0 * bytes / ms / 1e9);
    }
    CK(hipEventRecord(a)); CK(hipMemsetAsync(p, 0, bytes, 0)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemsetAsync 1 GiB: %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
    return 0;
}
