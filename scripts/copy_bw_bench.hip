// What is the float4-copy ceiling of THIS MI355X?  (scripts/write_bw_bench.hip's copy reached 5.12 TB/s read+write; the
// micro-architecture guide quotes 6.29 TB/s for a float4 copy.)  Sweep: block size, loads in flight per thread, grid size, nontemporal
// accesses, block-private contiguous segments, array size (256 MiB ... 2 GiB per array: the 256 MiB Infinity Cache absorbs small ones).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/copy_bw_bench.hip -o scripts/bin/copy_bw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void copy_gs(const v4 *__restrict__ s, v4 *__restrict__ d, size_t n) {   // grid-stride, U loads in flight
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&s[i + u * stride]) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &d[i + u * stride]); else d[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int U, bool NT>
__global__ void copy_seg(const v4 *__restrict__ s, v4 *__restrict__ d, size_t n_per_block) {   // every block copies its own contiguous segment
    const v4 *sb = s + static_cast<size_t>(blockIdx.x) * n_per_block;
    v4 *db = d + static_cast<size_t>(blockIdx.x) * n_per_block;
    size_t i = threadIdx.x;
    for (; i + (U - 1) * blockDim.x < n_per_block; i += U * blockDim.x) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&sb[i + u * blockDim.x]) : sb[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &db[i + u * blockDim.x]); else db[i + u * blockDim.x] = v[u]; }
    }
    for (; i < n_per_block; i += blockDim.x) db[i] = sb[i];
}
int main() {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    for (size_t mib : {256, 1024, 2048}) {
        const size_t bytes = mib << 20, n = bytes / 16;
        v4 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
        auto report = [&](const char *what) { printf("%5zu MiB  %-58s %8.1f us  %.2f TB/s read+write\n", mib, what, ms * 1e3, 2.0 * bytes / ms / 1e9); };
#define RUN(what, ...) do { float best = 1e9f; for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(a)); __VA_ARGS__; CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; } ms = best; report(what); } while (0)
        RUN("grid-stride, 1024 thr, 1 in flight, 512 blocks", hipLaunchKernelGGL((copy_gs<1, false>), dim3(512), dim3(1024), 0, 0, s, d, n));
        RUN("grid-stride, 256 thr, 4 in flight, 2048 blocks", hipLaunchKernelGGL((copy_gs<4, false>), dim3(2048), dim3(256), 0, 0, s, d, n));
        RUN("grid-stride, 256 thr, 8 in flight, 2048 blocks", hipLaunchKernelGGL((copy_gs<8, false>), dim3(2048), dim3(256), 0, 0, s, d, n));
        RUN("grid-stride, 256 thr, 8 in flight, 8192 blocks", hipLaunchKernelGGL((copy_gs<8, false>), dim3(8192), dim3(256), 0, 0, s, d, n));
        RUN("grid-stride, 512 thr, 4 in flight, 1024 blocks", hipLaunchKernelGGL((copy_gs<4, false>), dim3(1024), dim3(512), 0, 0, s, d, n));
        RUN("grid-stride, 256 thr, 4 in flight, 2048 blocks, nontemporal", hipLaunchKernelGGL((copy_gs<4, true>), dim3(2048), dim3(256), 0, 0, s, d, n));
        RUN("grid-stride, 256 thr, 8 in flight, 4096 blocks, nontemporal", hipLaunchKernelGGL((copy_gs<8, true>), dim3(4096), dim3(256), 0, 0, s, d, n));
        RUN("segments, 256 thr, 4 in flight, 2048 blocks", hipLaunchKernelGGL((copy_seg<4, false>), dim3(2048), dim3(256), 0, 0, s, d, n / 2048));
        RUN("segments, 256 thr, 8 in flight, 4096 blocks", hipLaunchKernelGGL((copy_seg<8, false>), dim3(4096), dim3(256), 0, 0, s, d, n / 4096));
        RUN("segments, 256 thr, 8 in flight, 16384 blocks", hipLaunchKernelGGL((copy_seg<8, false>), dim3(16384), dim3(256), 0, 0, s, d, n / 16384));
        RUN("segments, 256 thr, 8 in flight, 16384 blocks, nontemporal", hipLaunchKernelGGL((copy_seg<8, true>), dim3(16384), dim3(256), 0, 0, s, d, n / 16384));
        RUN("segments, 1024 thr, 4 in flight, 1024 blocks", hipLaunchKernelGGL((copy_seg<4, false>), dim3(1024), dim3(1024), 0, 0, s, d, n / 1024));
        RUN("hipMemcpyAsync device to device", CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)));
        CK(hipFree(s)); CK(hipFree(d));
    }
    return 0;
}
