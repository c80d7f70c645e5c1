// Stand-alone timing harness for k_hist_build (includes the kernel source directly so variants can be tried with -D flags).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gbrl_amd/csrc scripts/hist_bench.hip -o scripts/bin/hist_bench
#include "../gbrl_amd/csrc/kernels.hip"
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
    const int M = N / frac;
    std::vector<int32_t> rows(M);
    for (int i = 0; i < M; ++i) rows[i] = frac == 1 ? i : (i * frac + int(rng() % frac));
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
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hist_build(dc, N, dq, D, dr, dk, (int)chunks.size(), n_groups, FG, NB, dp, 0);
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
