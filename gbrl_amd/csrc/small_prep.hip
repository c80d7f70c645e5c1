// small_prep.hip -- RL-sized steps (round 5): the whole preparation of a step on ONE launch.
//
//   block f < F_pad   split candidates of numeric feature f and its class codes, straight from the caller's row-major matrix (no transposed
//                     key matrix): quantile candidates by sorting the column in LDS (sort_quantiles_body, <= 4096 rows), uniform candidates
//                     from the column's minimum / maximum (uniform_thresholds_body, <= 8192 rows); the padding features that fill the last
//                     group of 16 code slots write zeros
//   block F_pad       gradient statistics, fixed-point scales and quantised build gradients (small_stats_body), when the shape qualifies
//
// The level loop's preparation is five launches for such a step (k_small_stats, k_transpose_keys, k_sort_quantiles or k_column_minmax +
// k_uniform_thresholds + k_bin_cols, k_publish_pair), each a few microseconds of work behind a launch gap; the bodies are the SAME device
// functions (small_prep.h), so thresholds, codes, scales and quantised gradients keep their bits (tests/test_gpu_small_step.py runs both;
// GBRL_HIP_NO_SMALL_PREP=1 is the separate launches).
// Reference: Fitter::step_cpu (fitter.cpp:57-90), quantileSplitCandidates / uniformSplitCandidates (split_candidate_generator.cpp:59-115, 216-249).
#include "kernels.h"
#include "hooks.h"
#include "kernels_common.h"
#include "small_prep.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace gbrl {
namespace kern {

namespace {

struct SmallPrepArgs {
    const float *obs; int N, F, B, S, uniform, n_feat_blocks;
    const int64_t *cum; float *thr; uint32_t *thr_keys; uint16_t *codes, *codes_fm;
    const float *grads; int D, stat_blocks, stat_bs, centred, chunk_rows;
    double *stat; float *meanden; StepScales *sc; int32_t *qg;
    uint32_t *prof;
};

__global__ __launch_bounds__(kSmallStatsThreads) void k_small_prep(const SmallPrepArgs a) {
    extern __shared__ __align__(16) unsigned char prep_lds[];
    const int b = static_cast<int>(blockIdx.x);
    if (b == a.n_feat_blocks) {
        small_stats_body(a.grads, a.N, a.D, a.stat_blocks, a.stat_bs, a.centred, a.chunk_rows, a.stat, a.meanden, a.sc, a.qg, reinterpret_cast<double *>(prep_lds));
        return;
    }
    if (a.uniform) uniform_thresholds_body(a.obs, a.N, a.F, b, a.B, a.thr, a.thr_keys, a.codes, reinterpret_cast<uint32_t *>(prep_lds), a.codes_fm);
    else sort_quantiles_body<true>(a.obs, a.N, a.S, a.cum, a.B, a.thr_keys, a.thr, a.F, b, a.codes, reinterpret_cast<uint32_t *>(prep_lds), a.prof, a.codes_fm);
}

}  // namespace

bool small_stats_shape(int n, int D, int *n_blocks, int *bs, size_t *lds) {
    const int nb = column_sums_blocks(n, D);
    const int b = D <= 256 ? (256 / D) * D : D;
    if (D > 16 || nb * b > kSmallStatsVirtual || nb > 32 || n < 2) return false;
    const size_t l = sizeof(double) * (2 * static_cast<size_t>(nb) * b + static_cast<size_t>(nb) * 2 * D);
    if (l > 156 * 1024) return false;
    *n_blocks = nb; *bs = b; *lds = l;
    return true;
}

bool small_prep(const float *obs, int N, int F, int B, bool uniform, const int64_t *cum, float *thr, uint32_t *thr_keys, uint16_t *codes, uint16_t *codes_fm,
                const float *grads, int D, bool centred, int chunk_rows, double *stat, float *meanden, StepScales *sc, int32_t *qg,
                bool want_stats, bool *stats_done, hipStream_t s) {
    *stats_done = false;
    if (F < 1 || N < 1 || B < 1) return false;
    SmallPrepArgs a{};
    size_t lds = 0;
    if (uniform) {
        if (N > 8192) return false;
        lds = sizeof(uint32_t) * (static_cast<size_t>(N) + B + 32);
    } else {
        if (N > sort_quantiles_max_rows() || cum == nullptr) return false;
        int S = 256;
        while (S < N) S <<= 1;
        a.S = S;
        lds = sizeof(uint32_t) * (static_cast<size_t>(S) + B);
    }
    if (lds > 150 * 1024) return false;   // (the attribute below allows 158 KiB of dynamic LDS beside the static words)
    size_t stat_lds = 0;
    const bool with_stats = want_stats && small_stats_shape(N, D, &a.stat_blocks, &a.stat_bs, &stat_lds);
    lds = std::max(lds, with_stats ? stat_lds : static_cast<size_t>(0));
    static PerDeviceOnce attr;
    static uint64_t unsupported = 0;     // (remembered per device: a failed attribute call must not be retried as a launch with too much LDS)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr.first() && hipFuncSetAttribute(reinterpret_cast<const void *>(k_small_prep), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) != hipSuccess) {
        (void)hipGetLastError();
        if (dev >= 0 && dev < 64) unsupported |= 1ull << dev;
    }
    if (dev >= 0 && dev < 64 && ((unsupported >> dev) & 1ull)) return false;
    a.obs = obs; a.N = N; a.F = F; a.B = B; a.uniform = uniform ? 1 : 0;
    a.n_feat_blocks = ((F + kCodeGroup - 1) / kCodeGroup) * kCodeGroup;
    a.cum = cum; a.thr = thr; a.thr_keys = thr_keys; a.codes = codes; a.codes_fm = codes_fm;
    a.grads = grads; a.D = D; a.centred = centred ? 1 : 0; a.chunk_rows = chunk_rows; a.stat = stat; a.meanden = meanden; a.sc = sc; a.qg = qg;
    {   // measurement hook: GBRL_HIP_SMALL_PREP_PROF=1 prints feature 0's stage times of the previous launch
        static uint32_t *h_prof = nullptr;
        const bool on = [] { const char *e = hooks::raw(hooks::SMALL_PREP_PROF); return e && e[0] == '1'; }();
        if (on) {
            if (!h_prof) { (void)hipHostMalloc(reinterpret_cast<void **>(&h_prof), 64, hipHostMallocCoherent | hipHostMallocMapped); for (int i = 0; i < 16; ++i) h_prof[i] = 0; }
            else fprintf(stderr, "[small_prep feature 0, us] load %.2f sort %.2f thresholds %.2f codes %.2f\n", h_prof[0] / 100.0, h_prof[1] / 100.0, h_prof[2] / 100.0, h_prof[3] / 100.0);
            void *dp = nullptr;
            (void)hipHostGetDevicePointer(&dp, h_prof, 0);
            a.prof = static_cast<uint32_t *>(dp);
        }
    }
    hipLaunchKernelGGL(k_small_prep, dim3(a.n_feat_blocks + (with_stats ? 1 : 0)), dim3(kSmallStatsThreads), lds, s, a);
    if (hipGetLastError() != hipSuccess) return false;
    *stats_done = with_stats;
    return true;
}

}  // namespace kern
}  // namespace gbrl
