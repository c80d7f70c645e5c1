// Micro-benchmark: LDS atomic-add throughput on gfx950 for the access patterns the histogram kernel can choose from.
// hipcc --offload-arch=gfx950 -O3 scripts/lds_atomic_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 9;

// MODE 0: conflict-free (addr = (k*9+d)*64 + lane)            -> feature-fastest layout, 64 features
// MODE 1: random class, layout [(cls*9+d)*16 + f], 16 features x 4 rows per wave (2-way per 32-lane group)
// MODE 2: random class, layout [f][cls][d] (class-major per feature: random banks)
// MODE 3: like MODE 1 but 64-bit atomics (8 features)
// MODE 4/5: MODE 1 with a per-row-slot rotated field order that removes the bank conflicts between the rows of a wave
template <int MODE>
__global__ __launch_bounds__(1024) void bench(const uint32_t *__restrict__ codes, uint32_t *__restrict__ out, int nb) {
    extern __shared__ uint32_t h[];
    const int lds_words = (MODE == 3) ? nb * 9 * 8 * 2 : nb * 9 * 16;
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t seed = codes[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff];
    for (int it = 0; it < ITERS; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const int cls = (seed >> 8) % nb;
        if (MODE == 0) {
            const int base = ((cls % (nb / 4)) * 9) * 64 + lane;   // stays inside nb*9*16 words
#pragma unroll
            for (int d = 0; d < UNROLL; ++d) atomicAdd(&h[base + d * 64], seed >> 20);
        } else if (MODE == 1) {
            const int f = lane & 15;
            const int base = cls * 9 * 16 + f;
#pragma unroll
            for (int d = 0; d < UNROLL; ++d) atomicAdd(&h[base + d * 16], seed >> 20);
        } else if (MODE == 2) {
            const int f = lane & 15;
            const int base = (f * nb + cls) * 9;
#pragma unroll
            for (int d = 0; d < UNROLL; ++d) atomicAdd(&h[base + d], seed >> 20);
        } else if (MODE == 4 || MODE == 5) {
            // design layout, but every 16-lane row slot walks the 8 gradient fields in its own rotated order so that the row
            // slots of a wave always sit on different (class*9 + field) residues mod 4: field(t) = 4*(t>>2) + ((t + k) & 3),
            // k = (row slot - class) & 3.  MODE 5 also rotates the values (what the real kernel would have to do).
            const int f = lane & 15, k = ((lane >> 4) - cls) & 3;
            const int base = cls * 9 * 16 + f;
            uint32_t v[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) v[d] = (seed >> 20) + d;
            if (MODE == 5) {
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    uint32_t a0 = v[4 * hlf], a1 = v[4 * hlf + 1], a2 = v[4 * hlf + 2], a3 = v[4 * hlf + 3];
                    const bool r1 = k & 1, r2 = k & 2;
                    uint32_t b0 = r1 ? a1 : a0, b1 = r1 ? a2 : a1, b2 = r1 ? a3 : a2, b3 = r1 ? a0 : a3;
                    v[4 * hlf] = r2 ? b2 : b0; v[4 * hlf + 1] = r2 ? b3 : b1; v[4 * hlf + 2] = r2 ? b0 : b2; v[4 * hlf + 3] = r2 ? b1 : b3;
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) atomicAdd(&h[base + (4 * (t >> 2) + ((t + k) & 3)) * 16], v[t]);
            atomicAdd(&h[base + 8 * 16], 1u);
        } else {
            const int f = lane & 7;
            unsigned long long *h64 = reinterpret_cast<unsigned long long *>(h);
            const int base = cls * 9 * 8 + f;
#pragma unroll
            for (int d = 0; d < UNROLL; ++d) atomicAdd(&h64[base + d * 8], (unsigned long long)(seed >> 20));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = h[wave + 5];
}

template <int MODE>
int run(const char *name, int nb, size_t lds) {
    uint32_t *codes, *out;
    CHECK(hipMalloc(&codes, 65536 * 4));
    CHECK(hipMalloc(&out, 4096 * 4));
    std::vector<uint32_t> hc(65536);
    for (int i = 0; i < 65536; ++i) hc[i] = i * 2654435761u;
    CHECK(hipMemcpy(codes, hc.data(), 65536 * 4, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bench<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(1024), lds, 0, codes, out, nb);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(1024), lds, 0, codes, out, nb);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double atomics = double(blocks) * 1024 * ITERS * UNROLL;
    printf("%-52s %8.3f ms  %7.2f T atomics/s  (%.2f clk/wave-instr/CU @2.4GHz)\n", name, ms, atomics / ms / 1e9,
           2.4e9 * 256 / (atomics / 64 / (ms * 1e-3)));
    CHECK(hipFree(codes)); CHECK(hipFree(out));
    return 0;
}

int main() {
    const int nb = 257;
    if (run<0>("u32 conflict-free (lane-private banks)", nb, nb * 9 * 16 * 4)) return 1;
    if (run<1>("u32 [cls][d][16 feat] x 4 rows/wave (design layout)", nb, nb * 9 * 16 * 4)) return 1;
    if (run<2>("u32 [feat][cls][d] (random banks)", nb, nb * 9 * 16 * 4)) return 1;
    if (run<3>("u64 [cls][d][8 feat] x 8 rows/wave", nb, nb * 9 * 8 * 8)) return 1;
    if (run<4>("u32 design layout, rotated field order per row slot", nb, nb * 9 * 16 * 4)) return 1;
    if (run<5>("u32 ... + value rotation (8 v_cndmask per 4 fields)", nb, nb * 9 * 16 * 4)) return 1;
    return 0;
}
