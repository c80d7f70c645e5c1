// Micro-benchmark: no-return global (L2) atomic-add throughput on gfx950 for a histogram-like access pattern.
// hipcc --offload-arch=gfx950 -O3 scripts/global_atomic_bench.hip -o scripts/bin/gatom && scripts/bin/gatom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 256;
// each lane: feature fl = lane & 15 (16 features), random class; address = ((f*257 + cls)*9 + d)  [feature-major histogram]
template <int LAYOUT>
__global__ __launch_bounds__(1024) void bench(int32_t *__restrict__ hist, int n_feat) {
    uint32_t seed = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    const int fl = threadIdx.x & 15;
    const int fgroup = (blockIdx.x % (n_feat / 16));
    const int f = fgroup * 16 + fl;
    for (int it = 0; it < ITERS; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const int cls = (seed >> 8) % 257;
        int32_t *dst = LAYOUT == 0 ? hist + (static_cast<size_t>(f) * 257 + cls) * 9            // [f][cls][d]
                                   : hist + (static_cast<size_t>(fgroup) * 257 + cls) * 9 * 16 + fl;   // [group][cls][d][16]
#pragma unroll
        for (int d = 0; d < 9; ++d) atomicAdd(dst + (LAYOUT == 0 ? d : d * 16), static_cast<int32_t>(seed >> 20));
    }
}
template <int LAYOUT>
int run(const char *name) {
    const int n_feat = 128;
    int32_t *hist;
    CHECK(hipMalloc(&hist, sizeof(int32_t) * n_feat * 257 * 9 * 2));
    CHECK(hipMemset(hist, 0, sizeof(int32_t) * n_feat * 257 * 9 * 2));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(bench<LAYOUT>, dim3(blocks), dim3(1024), 0, 0, hist, n_feat);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(bench<LAYOUT>, dim3(blocks), dim3(1024), 0, 0, hist, n_feat);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double atomics = double(blocks) * 1024 * ITERS * 9;
    printf("%-40s %8.3f ms  %7.3f T atomics/s\n", name, ms, atomics / ms / 1e9);
    CHECK(hipFree(hist));
    return 0;
}
int main() {
    if (run<0>("global atomics [feature][class][d]")) return 1;
    if (run<1>("global atomics [group][class][d][16 f]")) return 1;
    return 0;
}
