// Microbenchmark for the "no-atomic" histogram idea: rows visited in class-sorted order, the row's
// gradient record gathered from LDS and summed in registers.  Measures, per CU, the clocks one wave-iteration (64 (row, feature)
// pairs) of the inner loop costs with all 16 waves of a block running it, for
//   mode 0: u16 index read (sorted list, contiguous per lane) + ds_read_b128 of a RANDOM 16-byte record (8 x int16) + 8 adds
//   mode 1: the same without the index read (record address computed: random but register-resident)
//   mode 2: records read at lane-linear addresses (conflict-free reference rate)
//   mode 3: 32-byte int32 records (two ds_read_b128, the form the verdict proposed), random
// against the 9 x ds_add_u32 per pair of the production kernel (4.46 clk each = 40 clk per 64 pairs, lds_atomic_bench.hip).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lds_gather_bench.hip -o scripts/bin/lds_gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int R = 2048, RUN = 32, REPS = 128;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint16_t *__restrict__ idx_g, int *__restrict__ out, long long *__restrict__ clk) {
    extern __shared__ int lds[];
    v4i *G = reinterpret_cast<v4i *>(lds);                                // R records of 16 B (mode 3: 32 B)
    constexpr int REC = MODE == 3 ? 32 : 16;
    uint16_t *sorted = reinterpret_cast<uint16_t *>(lds + R * REC / 4);   // [16][R] byte offsets
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < R * REC / 4; i += 1024) lds[i] = i * 2654435761u >> 20;
    for (int i = tid; i < 16 * R; i += 1024) sorted[i] = idx_g[i] * (REC / 16);
    __syncthreads();
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t0 = clock64();
    const char *Gb = reinterpret_cast<const char *>(G);
    for (int rep = 0; rep < REPS; ++rep) {
        asm volatile("" ::: "memory");     // the LDS contents are "new" every repetition: nothing is hoisted out of the loop
        const uint16_t *sp = sorted + wave * R + ((lane * RUN + rep * 32) & (R - 1));
        int o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = MODE == 0 || MODE == 3 ? sp[k] : 0;
        unsigned rnd = (lane * 2654435761u + rep * 40503u);
        for (int i = 0; i < RUN; i += 4) {
            v4i r[4], r2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int off;
                if (MODE == 0 || MODE == 3) off = o[k];
                else if (MODE == 1) { rnd = rnd * 1664525u + 1013904223u; off = (rnd >> 17) & (R * 16 - 16); }
                else off = ((lane + (i + k) * 64) * 16) & (R * 16 - 16);
                r[k] = *reinterpret_cast<const v4i *>(Gb + off);
                if (MODE == 3) r2[k] = *reinterpret_cast<const v4i *>(Gb + off + 16);
            }
            if (MODE == 0 || MODE == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = sp[(i + 4 + k) & (RUN - 1)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (MODE == 3) {
                    acc[0] += r[k].x; acc[1] += r[k].y; acc[2] += r[k].z; acc[3] += r[k].w;
                    acc[4] += r2[k].x; acc[5] += r2[k].y; acc[6] += r2[k].z; acc[7] += r2[k].w;
                } else {
                    acc[0] += static_cast<int16_t>(r[k].x & 0xffff); acc[1] += r[k].x >> 16;
                    acc[2] += static_cast<int16_t>(r[k].y & 0xffff); acc[3] += r[k].y >> 16;
                    acc[4] += static_cast<int16_t>(r[k].z & 0xffff); acc[5] += r[k].z >> 16;
                    acc[6] += static_cast<int16_t>(r[k].w & 0xffff); acc[7] += r[k].w >> 16;
                }
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    int s = 0;
    for (int d = 0; d < 8; ++d) s += acc[d];
    out[blockIdx.x * 1024 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const uint16_t *d_idx, int *d_out, long long *d_clk, const char *what) {
    const size_t lds = R * (MODE == 3 ? 32 : 16) + 16 * R * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, d_idx, d_out, d_clk);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> c(256);
    CK(hipMemcpy(c.data(), d_clk, 256 * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : c) mean += v; mean /= 256;
    const double iters = double(REPS) * RUN * 16;       // wave-iterations per CU (16 waves)
    printf("%-62s %7.1f us   %.1f clk per 64 pairs per CU (clock64), %.1f (event time at 2.3 GHz)\n", what, ms * 1e3, mean / iters,
           ms * 1e-3 * 2.3e9 / iters);
    return 0;
}
int main() {
    std::mt19937 rng(5);
    std::vector<uint16_t> idx(16 * R);
    for (int f = 0; f < 16; ++f) {
        std::vector<int> perm(R);
        for (int i = 0; i < R; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int i = 0; i < R; ++i) idx[f * R + i] = uint16_t(perm[i] * 16);
    }
    uint16_t *d_idx; int *d_out; long long *d_clk;
    CK(hipMalloc(&d_idx, idx.size() * 2)); CK(hipMalloc(&d_out, 256 * 1024 * 4)); CK(hipMalloc(&d_clk, 256 * 8));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * 2, hipMemcpyHostToDevice));
    if (run<0>(d_idx, d_out, d_clk, "mode 0: u16 index + random 16-B record (8 x int16) + 8 adds")) return 1;
    if (run<1>(d_idx, d_out, d_clk, "mode 1: random 16-B record, address from registers")) return 1;
    if (run<2>(d_idx, d_out, d_clk, "mode 2: lane-linear 16-B records (conflict-free reference)")) return 1;
    if (run<3>(d_idx, d_out, d_clk, "mode 3: u16 index + random 32-B record (8 x int32) + 8 adds")) return 1;
    printf("production k_hist_build: 9 ds_add_u32 x 4.46 clk = 40.1 clk per 64 pairs per CU (floor), 52.5 measured\n");
    return 0;
}
