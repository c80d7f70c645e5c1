// Stand-alone timing harness for k_hist_build (includes the kernel source directly).  The experiment variants of rounds 2-3 live
// HERE, as loader policies handed to the kernel's `Ld` template parameter (-DVARIANT=n), not in the product source:
//   0 (default) the product's launch;  1 no loads at all (class code and gradient from register arithmetic: the bare atomic loop);
//   2 non-temporal class-code loads;  3 16-bit gradient records (the harness then reads the int32 array as int16 pairs: timing only);
//   4 PAIR layout of the class codes (round 4): the 32-byte records of groups 2p and 2p+1 of a row share one 64-byte sector
//     ([pair][row][2][16] instead of [group][row][16]) -- timing only: the same buffer read at other addresses, modulo the class count
//   5 ROW-MAJOR layout (round 6): [row][128] uint16 -- a gathered row's eight 32-byte records are the two 128-byte lines of that row, shared by the
//     eight group blocks of a chunk through their XCD's L2, instead of one line per group with three other rows' records in it -- timing only
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gbrl_amd/csrc [-DVARIANT=n] scripts/hist_bench.hip gbrl_amd/csrc/hooks.cpp -o scripts/bin/hist_bench
#include "../gbrl_amd/csrc/kernels.hip"
#ifndef VARIANT
#define VARIANT 0
#endif
namespace gbrl { namespace kern {
struct ExpNoLoad {
    static __device__ __forceinline__ int code(const char *, uint32_t row, uint32_t coff) { return static_cast<int>((row * 2654435761u + coff * 40503u) >> 24); }
    template <int DT> static __device__ __forceinline__ int grad(const char *, uint32_t row, uint32_t qoff) { return static_cast<int>(row + qoff); }
};
struct ExpNtCodes : HistLoads {
    static __device__ __forceinline__ int code(const char *cgroup, uint32_t row, uint32_t coff) {
        return __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(cgroup + (row * (kCodeGroup * 2u) + coff)));
    }
};
struct ExpQ16 : HistLoads {
    template <int DT> static __device__ __forceinline__ int grad(const char *qbase, uint32_t row, uint32_t qoff) {
        return *reinterpret_cast<const int16_t *>(qbase + (row * static_cast<uint32_t>(DT * 2) + (qoff >> 1)));
    }
};
__device__ const char *g_codes_base;
__device__ uint32_t g_rows_total;
struct ExpPairLayout : HistLoads {
    static __device__ __forceinline__ int code(const char *cgroup, uint32_t row, uint32_t coff) {
        const uint32_t g = static_cast<uint32_t>((cgroup - g_codes_base) / (static_cast<size_t>(g_rows_total) * 32u));   // block-uniform
        const char *pair = g_codes_base + static_cast<size_t>(g >> 1) * g_rows_total * 64u + (g & 1u) * 32u;
        return *reinterpret_cast<const uint16_t *>(pair + (row * 64u + coff));
    }
};
struct ExpRowMajor : HistLoads {
    static __device__ __forceinline__ int code(const char *cgroup, uint32_t row, uint32_t coff) {
        const uint32_t g = static_cast<uint32_t>((cgroup - g_codes_base) / (static_cast<size_t>(g_rows_total) * 32u));   // block-uniform
        return *reinterpret_cast<const uint16_t *>(g_codes_base + (static_cast<size_t>(row) * 256u + g * 32u + coff));
    }
};
template <class Ld>
static void launch_variant(const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows, const Chunk *chunks, int n_chunks, int n_groups,
                           int NB, int32_t *partials) {
    auto k = k_hist_build<8, 8, true, Ld>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = static_cast<size_t>(NB) * (D + 1) * 16 * sizeof(int32_t);
    hipLaunchKernelGGL(k, dim3(8 * n_groups * ((n_chunks + 7) / 8)), dim3(kHistThreads), lds, 0, codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, 16, 4, NB, partials, HistDirect{});
}
}}
#include <cstdio>
#include <vector>
#include <random>
using namespace gbrl::kern;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    const int N = 1 << 20, F = 128, D = 8, NB = 257, FG = 16;
    const int frac = argc > 1 ? atoi(argv[1]) : 1;       // process N/frac rows (a gathered subset when frac > 1)
    const int chunk_rows = argc > 2 ? atoi(argv[2]) : 16384;
    const int n_groups = F / FG;
    std::mt19937 rng(1);
    std::vector<uint16_t> codes(size_t(N) * F);
    for (auto &c : codes) c = rng() % NB;
    std::vector<int32_t> qg(size_t(N) * D);
    for (auto &q : qg) q = int(rng() % 2001) - 1000;
    // argv[3] = 1: "level mode" -- N / 2 rows like a tree level below the root: frac / 2 nodes, node j = the rows of a random 1 / frac subset
    // (every row gets a random node id in [0, frac); nodes 0 .. frac/2 - 1 are accumulated), one node after the other in the row list
    const bool level_mode = argc > 3 && atoi(argv[3]) == 1 && frac >= 2;
    std::vector<int32_t> rows;
    if (level_mode) {
        std::vector<std::vector<int32_t>> node(frac);
        for (int i = 0; i < N; ++i) node[rng() % frac].push_back(i);
        for (int j = 0; j < frac / 2; ++j) rows.insert(rows.end(), node[j].begin(), node[j].end());
    } else {
        rows.resize(N / frac);
        for (size_t i = 0; i < rows.size(); ++i) rows[i] = frac == 1 ? int(i) : int(i * frac + rng() % frac);
    }
    const int M = static_cast<int>(rows.size());
    std::vector<Chunk> chunks;
    for (int off = 0; off < M; off += chunk_rows) chunks.push_back({0, off, std::min(chunk_rows, M - off), 0});
    uint16_t *dc; int32_t *dq, *dr, *dp; Chunk *dk;
    const size_t n_acc = size_t(NB) * (D + 1) * FG;
    CK(hipMalloc(&dc, codes.size() * 2)); CK(hipMalloc(&dq, qg.size() * 4)); CK(hipMalloc(&dr, rows.size() * 4));
    CK(hipMalloc(&dk, chunks.size() * sizeof(Chunk))); CK(hipMalloc(&dp, chunks.size() * n_groups * n_acc * 4));
    CK(hipMemcpy(dc, codes.data(), codes.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, qg.data(), qg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    { const char *cb = reinterpret_cast<const char *>(dc); uint32_t nn = N; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_codes_base), &cb, sizeof(cb))); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_total), &nn, sizeof(nn))); }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
#if VARIANT == 0
        hist_build(dc, N, dq, D, dr, dk, (int)chunks.size(), n_groups, FG, NB, dp, 0);
#elif VARIANT == 1
        launch_variant<ExpNoLoad>(dc, N, dq, D, dr, dk, (int)chunks.size(), n_groups, NB, dp);
#elif VARIANT == 2
        launch_variant<ExpNtCodes>(dc, N, dq, D, dr, dk, (int)chunks.size(), n_groups, NB, dp);
#elif VARIANT == 3
        launch_variant<ExpQ16>(dc, N, dq, D, dr, dk, (int)chunks.size(), n_groups, NB, dp);
#elif VARIANT == 4
        launch_variant<ExpPairLayout>(dc, N, dq, D, dr, dk, (int)chunks.size(), n_groups, NB, dp);
#else
        launch_variant<ExpRowMajor>(dc, N, dq, D, dr, dk, (int)chunks.size(), n_groups, NB, dp);
#endif
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double atomics = double(M) * F * (D + 1);
        printf("rows=%d chunks=%zu  %.1f us  %.2f T atomics/s  alg %.0f GB/s\n", M, chunks.size(), ms * 1e3, atomics / ms / 1e9,
               (double(M) * (F + D * 4 + 4)) / ms / 1e6);
    }
    // checksum
    std::vector<int32_t> hp(n_acc);
    CK(hipMemcpy(hp.data(), dp, n_acc * 4, hipMemcpyDeviceToHost));
    long long cs = 0; for (auto v : hp) cs += v;
    printf("checksum %lld\n", cs);
    return 0;
}
