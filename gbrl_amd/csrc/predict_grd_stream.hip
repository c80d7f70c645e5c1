// predict_grd_stream.hip -- gfx950 kernel behind GBRL::predict for SMALL GREEDY ensembles over LARGE batches (round 6; the configs[2] half of
// the headline metric: 2^20 x 128 rows through 10 depth-6 trees; reference semantics predictor.cpp:188-229 + optimizer.cpp:110-118).
//
// That regime is HBM-bound (570 MB of rows per call) and the block-cooperative kernel (k_predict_obl2<GREEDY>) leaves a third of the
// bandwidth unused: a greedy descent is a chain of 2 x depth dependent LDS reads per tree, its blocks hold their row tile AND wait at two
// barriers per tree group, and nothing is in flight from HBM while a block walks.  Here
//   * the whole ensemble (leaf values + node records, the mirror of predict_obl2) is staged in LDS ONCE per block -- the kernel is taken
//     only when it fits beside the row tiles (10 trees of 64 leaves x 8 outputs at 128 features);
//   * every WAVE is its own pipeline over 64-row tiles: no barrier after the staging.  A wave owns one LDS tile (row stride F + 1) and,
//     while it walks tile t, holds tile t + 1 IN REGISTERS (F / 4 float4 per lane, 128 VGPRs at 128 features: one wave per SIMD has 512) --
//     so every CU keeps 4 x 32 KB of row reads in flight the whole time, which is what 1/256 of 5.5 TB/s needs at ~2 us of loaded latency;
//   * lane = row: all outputs of a row accumulate in the lane's registers, p = fma(-lr, v, p) tree by tree -- the operation sequence of
//     k_predict_obl2 / the general kernel, so the bits are theirs (tests/test_gpu_predict_grd_stream.py, scripts/grd_stream_sweep.py compare the paths);
//   * four trees descend together (independent chains hide the LDS latency that one wave per SIMD cannot hide otherwise).
#include "kernels.h"
#include "hooks.h"
#include "kernels_common.h"

#include <algorithm>
#include <cstdlib>

namespace gbrl {
namespace kern {

namespace {

// Waves per block (= per CU): four -- one per SIMD -- when the ensemble leaves room for four tiles, three for larger ensembles (11-21 trees at 128
// features: 133-193 us against 149-209 for the cooperative kernel); with two the walk no longer overlaps enough (GBRL_HIP_PREDICT_GRD_STREAM_WAVES=2).
constexpr size_t kGsLds = 160 * 1024;   // bytes of LDS a block may ask for

template <int DMAX>
struct GsCoef { float lr[DMAX]; };

template <int MAXD, int DMAX, int NV, int kGsWaves>
__global__ __launch_bounds__(64 * kGsWaves) void k_predict_grd_stream(const float *__restrict__ vsw, const float *__restrict__ bias, GsCoef<DMAX> coef, int D,
                                                                       const float *__restrict__ obs, int n, int start_tree, int trees,
                                                                       float *__restrict__ out, int n_tiles) {
    extern __shared__ float lds[];
    constexpr int W = 4, DW = DMAX / W, LS = 1 << MAXD, VT = LS * DMAX, REC = VT * 4 + LS * 16, F = 4 * NV, XS = F + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *recs = reinterpret_cast<char *>(lds);
    float *tile = reinterpret_cast<float *>(recs + static_cast<size_t>(trees) * REC) + wave * 64 * XS;
    float4 R[NV];     // the NEXT tile, on its way from HBM while the current one is walked
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        const float4 *src = reinterpret_cast<const float4 *>(obs + static_cast<size_t>(t) * 64 * F);
        const int last = min(64, n - t * 64) * NV - 1;      // the last tile may be short: its lanes re-read the last valid piece
#pragma unroll
        for (int i = 0; i < NV; ++i) R[i] = src[min(i * 64 + lane, last)];
    };
    const int stride = gridDim.x * kGsWaves;
    int t = blockIdx.x * kGsWaves + wave;
    if (t < n_tiles) load_tile(t);      // (first: it is in flight while the records are staged)
    {   // the ensemble's records: a straight 16-byte copy of the mirror (padded: whole float4s exist), eight pieces per thread in flight
        const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(vsw) + static_cast<size_t>(start_tree) * REC);
        float4 *dst = reinterpret_cast<float4 *>(recs);
        const int n16 = trees * (REC / 16);
        constexpr int NT = 64 * kGsWaves;
        for (int i0 = tid; i0 < n16; i0 += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i0 + u * NT];      // (beyond n16: the mirror's padding)
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * NT < n16) dst[i0 + u * NT] = v[u];
        }
    }
    float nlr[DMAX], p0[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) {
        nlr[j] = -coef.lr[j];
        p0[j] = j < D ? 0.0f + bias[j] : 0.0f;
    }
    __syncthreads();
    const float *x = tile + lane * XS;
    while (t < n_tiles) {
        // registers -> this wave's LDS tile (piece g of the tile = row g / NV, features 4 (g % NV) ..)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int g = i * 64 + lane;
            float *dst = tile + (g / NV) * XS + (g % NV) * 4;
            dst[0] = R[i].x; dst[1] = R[i].y; dst[2] = R[i].z; dst[3] = R[i].w;
        }
        const int next = t + stride;
        if (next < n_tiles) load_tile(next);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();     // (one wave: its LDS operations complete in order; this keeps the compiler from moving reads up)
        float p[DMAX];
#pragma unroll
        for (int j = 0; j < DMAX; ++j) p[j] = p0[j];
        auto walk = [&](int t0, auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            const char *nb = recs + static_cast<size_t>(t0) * REC + VT * 4;
            int node[K];
#pragma unroll
            for (int k = 0; k < K; ++k) node[k] = 0;
#pragma unroll
            for (int d = 0; d < MAXD; ++d) {
                int4 rec[K];
                float xv[K];
#pragma unroll
                for (int k = 0; k < K; ++k) rec[k] = *reinterpret_cast<const int4 *>(nb + k * REC + max(node[k], 0) * 16);
#pragma unroll
                for (int k = 0; k < K; ++k) xv[k] = x[rec[k].x];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool right = xv[k] > __int_as_float(rec[k].y);     // false for NaN, like the reference's
                    const int nxt = right ? rec[k].w : rec[k].z;
                    node[k] = node[k] >= 0 ? nxt : node[k];
                }
            }
            // leaf values [worker w][leaf][DW] (the mirror's layout), applied in tree order
            float v[K][DMAX];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const char *vb = recs + static_cast<size_t>(t0 + k) * REC + static_cast<uint32_t>(~node[k]) * (DW * 4);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const float *vv = reinterpret_cast<const float *>(vb + w * LS * DW * 4);
                    if constexpr (DW == 2) {
                        const float2 t2 = *reinterpret_cast<const float2 *>(vv);
                        v[k][w * DW] = t2.x; v[k][w * DW + 1] = t2.y;
                    } else {
#pragma unroll
                        for (int jj = 0; jj < DW; ++jj) v[k][w * DW + jj] = vv[jj];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < DMAX; ++j) p[j] = fmaf(nlr[j], v[k][j], p[j]);
        };
        int tr = 0;
        for (; tr + 4 <= trees; tr += 4) walk(tr, std::integral_constant<int, 4>{});
        for (; tr < trees; ++tr) walk(tr, std::integral_constant<int, 1>{});
        const int row = t * 64 + lane;
        if (row < n) {
            float *o = out + static_cast<size_t>(row) * D;
            if (D == DMAX) {
#pragma unroll
                for (int c = 0; c < DMAX / 4; ++c) *reinterpret_cast<float4 *>(o + 4 * c) = make_float4(p[4 * c], p[4 * c + 1], p[4 * c + 2], p[4 * c + 3]);
            } else {
#pragma unroll
                for (int j = 0; j < DMAX; ++j) if (j < D) o[j] = p[j];
            }
        }
        __builtin_amdgcn_wave_barrier();     // the tile is rewritten next: its reads are done (same wave, in order)
        t = next;
    }
}

template <int MAXD, int DMAX, int NV, int kGsWaves>
bool launch_gs_w(const PredictModel &pm, const float *obs, int n, int start_tree, int stop_tree, float *out, hipStream_t s) {
    constexpr int LS = 1 << MAXD, REC = LS * DMAX * 4 + LS * 16, F = 4 * NV, XS = F + 1;
    const int trees = stop_tree - start_tree;
    const size_t lds = static_cast<size_t>(trees) * REC + static_cast<size_t>(kGsWaves) * 64 * XS * 4;
    if (lds > kGsLds) return false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static uint64_t attr_done = 0;   // per device
    static int cus[64] = {0};
    if (dev < 64 && !((attr_done >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_predict_grd_stream<MAXD, DMAX, NV, kGsWaves>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kGsLds));
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev] = c;
        attr_done |= 1ull << dev;
    }
    GsCoef<DMAX> coef;
    for (int j = 0; j < DMAX; ++j) coef.lr[j] = j < pm.D ? pm.coef[j] : 0.0f;
    const int n_tiles = (n + 63) / 64;
    const int blocks = std::min(dev < 64 ? cus[dev] : 256, (n_tiles + kGsWaves - 1) / kGsWaves);
    hipLaunchKernelGGL((k_predict_grd_stream<MAXD, DMAX, NV, kGsWaves>), dim3(blocks), dim3(64 * kGsWaves), lds, s, pm.values_sw, pm.bias, coef, pm.D, obs, n, start_tree, trees,
                       out, n_tiles);
    return true;
}

template <int MAXD, int DMAX, int NV>
bool launch_gs(const PredictModel &pm, const float *obs, int n, int start_tree, int stop_tree, float *out, hipStream_t s) {
    int min_waves = 3;      // (two waves: 264 against 215 us for the cooperative kernel at 24 trees -- scripts/greedy_predict_sweep.py)
    if (const char *e = hooks::raw(hooks::PREDICT_GRD_STREAM_WAVES)) min_waves = std::max(1, std::atoi(e));      // measurement hook: 4 = only the four-wave shape
    return launch_gs_w<MAXD, DMAX, NV, 4>(pm, obs, n, start_tree, stop_tree, out, s) ||
           (min_waves <= 3 && launch_gs_w<MAXD, DMAX, NV, 3>(pm, obs, n, start_tree, stop_tree, out, s)) ||
           (min_waves <= 2 && launch_gs_w<MAXD, DMAX, NV, 2>(pm, obs, n, start_tree, stop_tree, out, s));
}

template <int MAXD, int DMAX>
bool launch_gs_f(const PredictModel &pm, const float *obs, int F, int n, int start_tree, int stop_tree, float *out, hipStream_t s) {
    switch (F) {
        case 32: return launch_gs<MAXD, DMAX, 8>(pm, obs, n, start_tree, stop_tree, out, s);
        case 64: return launch_gs<MAXD, DMAX, 16>(pm, obs, n, start_tree, stop_tree, out, s);
        case 96: return launch_gs<MAXD, DMAX, 24>(pm, obs, n, start_tree, stop_tree, out, s);
        case 128: return launch_gs<MAXD, DMAX, 32>(pm, obs, n, start_tree, stop_tree, out, s);
        default: return false;
    }
}

}  // namespace

// Greedy ensembles whose records fit in LDS beside four (or three) 64-row tiles, numeric rows of 32 / 64 / 96 / 128 features, at most 8 outputs,
// at least 2^18 rows: false = not covered, nothing was launched.
bool predict_grd_stream(const PredictModel &pm, const float *obs, int F, int Fc, int n, int start_tree, int stop_tree, float *out, hipStream_t s) {
    // from 2^18 rows on (four tiles per wave): below that the pipeline has too few tiles per wave to pay for its start -- 32 768 rows 11.4 against
    // 8.6 us for the cooperative kernel, 2^17 23.0 / 21.8, 2^18 34.5 / 34.6, 2^19 70 / 81, 2^20 111 / 145 (GBRL_HIP_PREDICT_GRD_STREAM_MIN_ROWS: tests)
    const int min_rows = hooks::num(hooks::PREDICT_GRD_STREAM_MIN_ROWS, 1 << 18);
    if (pm.oblivious || !pm.grd_ok || pm.values_sw == nullptr || Fc > 0 || pm.tree_chunk != 0 || n < min_rows || stop_tree <= start_tree) return false;
    if (hooks::on(hooks::PREDICT_NO_GRD_STREAM)) return false;      // test / measurement hook: the block-cooperative kernel (same bits)
    // a hook that selects the cooperative kernel's launch plan (or the first-generation kernel) is set: that kernel is what the caller wants to run
    if (hooks::on(hooks::PREDICT_OBL1) || hooks::on(hooks::PREDICT_NO_PERSIST) || hooks::raw(hooks::PREDICT_RG) || hooks::raw(hooks::PREDICT_TT) || hooks::raw(hooks::PREDICT_NB))
        return false;
    if ((reinterpret_cast<uintptr_t>(obs) & 15) != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0) return false;
    const int DMAX = obl2_padded_outputs(pm.D);
    if (pm.obl2_maxd == 6 && DMAX == 8) return launch_gs_f<6, 8>(pm, obs, F, n, start_tree, stop_tree, out, s);
    if (pm.obl2_maxd == 6 && DMAX == 4) return launch_gs_f<6, 4>(pm, obs, F, n, start_tree, stop_tree, out, s);
    if (pm.obl2_maxd == 4 && DMAX == 8) return launch_gs_f<4, 8>(pm, obs, F, n, start_tree, stop_tree, out, s);
    if (pm.obl2_maxd == 4 && DMAX == 4) return launch_gs_f<4, 4>(pm, obs, F, n, start_tree, stop_tree, out, s);
    return false;
}

}  // namespace kern
}  // namespace gbrl
