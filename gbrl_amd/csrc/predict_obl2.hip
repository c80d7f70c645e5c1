// predict_obl2.hip -- gfx950 kernel behind GBRL::predict for oblivious ensembles (A13: predictor.cpp:231-265 + optimizer.cpp:110-118),
// second generation.  The first-generation kernel (k_predict_obl, predict.hip) keeps one wave per SIMD because a block's row
// tile fills the LDS, so nothing hides its LDS / scalar-load latencies (measured: ~1800 cycles per (wave, tree)).  This kernel
// keeps the reference's per-row, per-output accumulation chain in TREE ORDER (Q14) and still runs 3-4 waves per SIMD:
//
//   * a block owns R = 64*RG rows (tile staged once in LDS, row stride odd) and runs 4 waves ("workers") per group of 64 rows;
//   * phase A (order-free): worker q finds the leaf of its row in trees q*TT/4 .. (q+1)*TT/4 of the current group of TT trees
//     -- the tree's (feature, threshold) pairs arrive through the scalar cache from a right-aligned record (shallower trees
//     are padded in front with a never-true condition, so every tree costs MAXD branch-free levels) -- and publishes the leaf
//     indices, four to a dword, in LDS;
//   * phase B (ordered): worker q owns outputs [q*DW, (q+1)*DW) of the row: for every tree of the group IN ORDER it reads the
//     leaf index, the leaf's DW values (one LDS read) and applies p = fma(-lr, v, p) -- the same operation sequence per output
//     as the reference's loop, so results are bitwise those of the general kernel;
//   * leaf values come from a pre-swizzled mirror ([tree][worker][leaf][DW], zero padded), so staging a group is a straight
//     16-byte copy; groups are double-buffered (values of group g+1 are written and those of g+2 are in flight while group g is
//     consumed) with ONE barrier per group (or single-buffered with two, when that buys a fourth wave per SIMD).
//   * categorical conditions compare 16-bit dictionary ids packed behind the row's numeric features in the same LDS tile.
#include "kernels.h"
#include "kernels_common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace gbrl {
namespace kern {

namespace {

constexpr int kObl2Workers = 4;
constexpr int kObl2MaxVec = 4;   // float4 registers per thread that carry a group's values from global memory to LDS

template <int DMAX>
struct Obl2Coef { float lr[DMAX]; };

// leaf = 2 * leaf + (x > t): the compare writes VCC, the add-with-carry doubles the index and takes the bit (2 VALU per level
// instead of compare + select + shift/or).  `t` is wave-uniform (scalar register).  x > t is false for NaN, like the reference's.
__device__ __forceinline__ void push_gt(uint32_t &leaf, float x, float t) {
    asm("v_cmp_lt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(leaf) : "v"(x), "s"(t) : "vcc");
}
__device__ __forceinline__ void push_eq(uint32_t &leaf, uint32_t code, uint32_t id) {
    asm("v_cmp_eq_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(leaf) : "v"(code), "s"(id) : "vcc");
}

// LDS map (bytes from 0): leaf values vt[NB][TT][W][LS][DW] f32 (below 64 KiB: their byte offsets travel as 16-bit fields),
// leaf offsets idx[2][W][NW][R] u32, row tile xt[R][xs] f32.
template <int DMAX, int MAXD, bool CAT>
__global__ __launch_bounds__(DMAX >= 64 ? 512 : 1024) void k_predict_obl2(const float *__restrict__ vsw, const int32_t *__restrict__ cond,
                                                       const float *__restrict__ bias, Obl2Coef<DMAX> coef, int D,
                                                       const float *__restrict__ obs, int F, const int32_t *__restrict__ cat_codes, int Fc,
                                                       int n, int start_tree, int stop_tree, float *__restrict__ out, int R, int TT, int NB,
                                                       int xs, int tree_chunk) {
    extern __shared__ float lds[];
    constexpr int W = kObl2Workers, DW = DMAX / W, LS = 1 << MAXD, VT = LS * DMAX;
    if (tree_chunk > 0) {   // small batches: this block column covers a sub-range of the trees and writes a partial sum (no bias)
        start_tree += blockIdx.y * tree_chunk;
        stop_tree = min(stop_tree, start_tree + tree_chunk);
        out += static_cast<size_t>(blockIdx.y) * n * D;
    }
    const int NT = blockDim.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave / W, q = wave % W;
    const int TPW = TT / W, NW = TPW > 2 ? 2 : 1;
    float *vt = lds;
    uint32_t *idx = reinterpret_cast<uint32_t *>(vt + static_cast<size_t>(NB) * TT * VT);
    float *xt = reinterpret_cast<float *>(idx + 2 * W * NW * R);
    const int r0 = blockIdx.x * R;
    const int rows = min(R, n - r0);
    const int row_l = rg * 64 + lane;
    const bool live = row_l < rows;
    const int n_groups = (stop_tree - start_tree + TT - 1) / TT;

    // A group's values travel global -> registers (kObl2MaxVec = 4 float4 per thread) -> LDS.  The loads are clamped, not predicated
    // (a predicated load becomes a branch with a wait behind every load): the four must be in flight together, and stay in
    // flight -- they are consumed one barrier later (store_vals).  Named registers: an array here is demoted to scratch.
    float4 va, vb, vc, vd;
    static_assert(kObl2MaxVec == 4, "four named registers below");
    auto load_vals = [&](int g) __attribute__((always_inline)) {
        const int t0 = start_tree + g * TT, tn = min(TT, stop_tree - t0);
        const float4 *src = reinterpret_cast<const float4 *>(vsw + static_cast<size_t>(t0) * VT);
        const int last = tn * (VT >> 2) - 1;
        va = src[min(tid, last)];
        vb = src[min(NT + tid, last)];
        vc = src[min(2 * NT + tid, last)];
        vd = src[min(3 * NT + tid, last)];
    };
    auto store_vals = [&](int g) __attribute__((always_inline)) {
        const int t0 = start_tree + g * TT, tn = min(TT, stop_tree - t0);
        float4 *dst = reinterpret_cast<float4 *>(vt + static_cast<size_t>(NB == 2 ? (g & 1) : 0) * TT * VT);
        const int cnt4 = tn * (VT >> 2);
        if (tid < cnt4) dst[tid] = va;
        if (NT + tid < cnt4) dst[NT + tid] = vb;
        if (2 * NT + tid < cnt4) dst[2 * NT + tid] = vc;
        if (3 * NT + tid < cnt4) dst[3 * NT + tid] = vd;
    };
    const float *x = xt + row_l * xs;
    // phase A: byte offsets (leaf * DW * 4) of this row's leaves in this worker's K trees of group g, 16 bits each ->
    // idx[g & 1][q][0..NW)[row].  The K records arrive together through the scalar cache (the mirror is padded, so trees past the
    // end of the range in the last group read valid records; their leaves are never applied).
    auto phase_a_k = [&](int g, auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        constexpr int NWK = K > 2 ? 2 : 1;
        // records of a batch of trees sit in scalar registers together: at most ~36 words (more spills SGPRs into VGPR lanes)
        constexpr int SBMAX = MAXD <= 4 ? 4 : MAXD <= 6 ? 3 : 2;
        constexpr int SB = K <= SBMAX ? K : (K + 1) / 2;
        const int32_t *cp0 = cond + static_cast<size_t>(start_tree + g * TT + q * K) * 2 * MAXD;
        uint32_t off[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k0 = 0; k0 < K; k0 += SB) {
            int fi[SB][MAXD];
            float tv[SB][MAXD], xv[SB][MAXD];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                if (k0 + k < K) {
                    const int32_t *cp = cp0 + (k0 + k) * 2 * MAXD;
#pragma unroll
                    for (int d = 0; d < MAXD; ++d) { fi[k][d] = cp[2 * d]; tv[k][d] = __int_as_float(cp[2 * d + 1]); }
                }
            }
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                if (k0 + k < K) {
#pragma unroll
                    for (int d = 0; d < MAXD; ++d) {
                        if (!CAT || fi[k][d] >= 0) xv[k][d] = x[fi[k][d]];
                        else xv[k][d] = x[F + ((~fi[k][d]) >> 1)];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                if (k0 + k < K) {
                    uint32_t leaf = 0;
#pragma unroll
                    for (int d = 0; d < MAXD; ++d) {
                        if (!CAT || fi[k][d] >= 0) {
                            push_gt(leaf, xv[k][d], tv[k][d]);
                        } else {
                            const uint32_t w = __float_as_uint(xv[k][d]);
                            const uint32_t code = ((~fi[k][d]) & 1) ? (w >> 16) : (w & 0xffffu);
                            push_eq(leaf, code, __float_as_uint(tv[k][d]));
                        }
                    }
                    off[k0 + k] = leaf * (DW * 4);
                }
            }
        }
        uint32_t *ib = idx + (((g & 1) * W + q) * NW) * R + row_l;
        ib[0] = off[0] | (off[1] << 16);
        if (NWK == 2) ib[R] = off[2] | (off[3] << 16);
    };
    auto phase_a = [&](int g) __attribute__((always_inline)) {
        switch (TPW) {
            case 1: phase_a_k(g, std::integral_constant<int, 1>{}); break;
            case 2: phase_a_k(g, std::integral_constant<int, 2>{}); break;
            case 3: phase_a_k(g, std::integral_constant<int, 3>{}); break;
            default: phase_a_k(g, std::integral_constant<int, 4>{}); break;
        }
    };
    float p[DW], lr[DW];
#pragma unroll
    for (int jj = 0; jj < DW; ++jj) {
        // outputs of this worker; the learning rates are selected with scalar compares (a dynamically indexed by-value struct
        // would be copied to scratch memory)
        float l = coef.lr[jj];
#pragma unroll
        for (int w = 1; w < W; ++w) l = q == w ? coef.lr[w * DW + jj] : l;
        lr[jj] = l;
        const int j = q * DW + jj;
        p[jj] = (j < D && tree_chunk == 0) ? 0.0f + bias[j] : 0.0f;
    }
    // phase B: apply the trees of group g in order to this worker's outputs.  K = trees per worker.  The 16-bit fields become
    // complete LDS addresses with one packed add per word (this worker's slice, the buffer); the tree's base is the read's
    // immediate offset.  The reads of a worker's K trees are in flight together, the fused multiply-adds follow in tree order.
    auto phase_b_k = [&](int g, auto kc, auto fullc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        constexpr int NWK = K > 2 ? 2 : 1;
        constexpr bool FULL = decltype(fullc)::value;
        const int t0 = start_tree + g * TT, tn = min(TT, stop_tree - t0);
        const uint32_t *ib = idx + static_cast<size_t>(g & 1) * W * NW * R + row_l;
        const uint32_t qoff = static_cast<uint32_t>(((NB == 2 ? (g & 1) : 0) * TT * VT + q * LS * DW) * 4);
        const uint32_t qoff2 = qoff | (qoff << 16);
        uint32_t word[W][NWK];
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
            for (int i = 0; i < NWK; ++i) word[w][i] = ib[(w * NW + i) * R] + qoff2;
        const char *lb = reinterpret_cast<const char *>(lds);
        constexpr int KB = DW >= 16 ? 1 : (DW >= 8 && K > 2) ? 2 : K;   // value registers in flight: KB * DW
#pragma unroll
        for (int w = 0; w < W; ++w) {
#pragma unroll
            for (int k0 = 0; k0 < K; k0 += KB) {
                float vv[KB][DW];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const int k = k0 + kk;
                    if (k < K) {
                        const uint32_t wd = word[w][k >> 1];
                        const uint32_t a = (k & 1) ? (wd >> 16) : (wd & 0xffffu);
                        const float *v = reinterpret_cast<const float *>(lb + a) + (w * K + k) * VT;
                        if (DW == 1) {
                            vv[kk][0] = v[0];
                        } else if (DW == 2) {
                            const float2 t2 = *reinterpret_cast<const float2 *>(v);
                            vv[kk][0] = t2.x; vv[kk][1] = t2.y;
                        } else {
#pragma unroll
                            for (int c = 0; c < DW / 4; ++c) {
                                const float4 t4 = *reinterpret_cast<const float4 *>(v + 4 * c);
                                vv[kk][4 * c] = t4.x; vv[kk][4 * c + 1] = t4.y; vv[kk][4 * c + 2] = t4.z; vv[kk][4 * c + 3] = t4.w;
                            }
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const int k = k0 + kk;
                    if (k < K && (FULL || w * K + k < tn)) {   // uniform
#pragma unroll
                        for (int jj = 0; jj < DW; ++jj) p[jj] = fmaf(-lr[jj], vv[kk][jj], p[jj]);
                    }
                }
            }
        }
    };
    auto phase_b = [&](int g) __attribute__((always_inline)) {
        const bool full = start_tree + (g + 1) * TT <= stop_tree;
        if (full) {
            switch (TPW) {
                case 1: phase_b_k(g, std::integral_constant<int, 1>{}, std::true_type{}); break;
                case 2: phase_b_k(g, std::integral_constant<int, 2>{}, std::true_type{}); break;
                case 3: phase_b_k(g, std::integral_constant<int, 3>{}, std::true_type{}); break;
                default: phase_b_k(g, std::integral_constant<int, 4>{}, std::true_type{}); break;
            }
        } else {
            switch (TPW) {
                case 1: phase_b_k(g, std::integral_constant<int, 1>{}, std::false_type{}); break;
                case 2: phase_b_k(g, std::integral_constant<int, 2>{}, std::false_type{}); break;
                case 3: phase_b_k(g, std::integral_constant<int, 3>{}, std::false_type{}); break;
                default: phase_b_k(g, std::integral_constant<int, 4>{}, std::false_type{}); break;
            }
        }
    };

    if (n_groups > 0) load_vals(0);
    // row tile: numeric features (coalesced 16-byte reads when F % 4 == 0), then the packed categorical ids
    {
        const float *src = obs + static_cast<size_t>(r0) * F;
        if (F > 0 && (F & 3) == 0) {
            const float4 *src4 = reinterpret_cast<const float4 *>(src);
            const int F4 = F >> 2, tot4 = rows * F4;
            constexpr int UL = 4;
            for (int i0 = tid; i0 < tot4; i0 += NT * UL) {
                float4 v[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const int i = i0 + u * NT;
                    v[u] = i < tot4 ? src4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const int i = i0 + u * NT;
                    if (i < tot4) {
                        const int r = i / F4, f = (i - r * F4) << 2;
                        float *dst = xt + r * xs + f;
                        dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
                    }
                }
            }
        } else {
            const int tot = rows * F;
            for (int i = tid; i < tot; i += NT) {
                const int r = i / F, f = i - r * F;
                xt[r * xs + f] = src[i];
            }
        }
        if (CAT) {
            const int FC2 = (Fc + 1) >> 1, tot = rows * FC2;
            const int32_t *cs = cat_codes + static_cast<size_t>(r0) * Fc;
            for (int i = tid; i < tot; i += NT) {
                const int r = i / FC2, c = i - r * FC2;
                const uint32_t lo = static_cast<uint32_t>(cs[r * Fc + 2 * c]) & 0xffffu;
                const uint32_t hi = (2 * c + 1 < Fc) ? (static_cast<uint32_t>(cs[r * Fc + 2 * c + 1]) & 0xffffu) : 0u;
                xt[r * xs + F + c] = __uint_as_float(lo | (hi << 16));
            }
        }
        // rows beyond the batch (last block): their lanes still walk the trees; give them defined words
        for (int i = rows * xs + tid; i < R * xs; i += NT) xt[i] = 0.0f;
    }
    if (n_groups > 0) {
        store_vals(0);
        load_vals(min(1, n_groups - 1));
    }
    __syncthreads();
    if (n_groups > 0) phase_a(0);
    // The prefetch loads are unconditional (group index clamped): a load under a condition ends in a register copy behind a
    // full wait at the join, which would expose the global-memory latency every group.
    for (int g = 0; g < n_groups; ++g) {
        __syncthreads();   // offsets and values of group g are published; group g-1 is fully consumed
        if (NB == 2) {
            if (g + 1 < n_groups) store_vals(g + 1);
            load_vals(min(g + 2, n_groups - 1));
            if (g + 1 < n_groups) phase_a(g + 1);
            phase_b(g);
        } else {   // one value buffer: the next group's values replace this group's after everybody has applied them
            if (g + 1 < n_groups) phase_a(g + 1);
            phase_b(g);
            __syncthreads();
            if (g + 1 < n_groups) store_vals(g + 1);
            load_vals(min(g + 2, n_groups - 1));
        }
    }
    if (live) {
        float *o = out + static_cast<size_t>(r0 + row_l) * D + q * DW;
#pragma unroll
        for (int jj = 0; jj < DW; ++jj)
            if (q * DW + jj < D) o[jj] = p[jj];
    }
}

struct Obl2Plan { int RG, TT, NB, xs; size_t lds; };

// Rows per block (64 * RG), trees per group (TT) and value buffers (NB): as many waves per CU as the LDS allows while a group
// still holds >= 8 trees; small ensembles take 64-row blocks so that several blocks per CU overlap their tile loads with each
// other's walks.  The value buffers must lie below 64 KiB (16-bit offsets).
static bool obl2_plan(int F, int Fc, bool cat, int maxd, int DMAX, int trees, Obl2Plan &pl) {
    const int rg_max = DMAX >= 64 ? 2 : 4;   // 16 accumulators + 16 rates per thread: that width is compiled for 512-thread blocks
    const size_t budget = 160 * 1024 - 256;
    const size_t vtb = (static_cast<size_t>(1) << maxd) * DMAX * sizeof(float);
    pl.xs = (F + (cat ? (Fc + 1) / 2 : 0)) | 1;
    auto lds_for = [&](int rg, int tt, int nb) -> size_t {
        const int nw = tt / kObl2Workers > 2 ? 2 : 1;
        return static_cast<size_t>(nb) * tt * vtb + static_cast<size_t>(2) * kObl2Workers * nw * 64 * rg * 4 + static_cast<size_t>(64) * rg * pl.xs * 4;
    };
    auto tt_for = [&](int rg, int nb) -> int {
        int tt = 16;
        while (tt >= 4 && (lds_for(rg, tt, nb) > budget || static_cast<size_t>(nb) * tt * vtb > 65536 ||
                           static_cast<size_t>(tt) * vtb / 16 > static_cast<size_t>(kObl2MaxVec) * 256 * rg)) tt -= 4;
        return tt >= 4 ? tt : 0;
    };
    int rg_env = 0, tt_env = 0, nb_env = 0;
    if (const char *e = std::getenv("GBRL_HIP_PREDICT_RG")) rg_env = std::atoi(e);
    if (const char *e = std::getenv("GBRL_HIP_PREDICT_TT")) tt_env = std::atoi(e);
    if (const char *e = std::getenv("GBRL_HIP_PREDICT_NB")) nb_env = std::atoi(e);
    int best_rg = 0, best_tt = 0, best_nb = 2;
    if (rg_env >= 1 && rg_env <= rg_max) {
        best_rg = rg_env; best_nb = nb_env == 1 ? 1 : 2; best_tt = tt_for(rg_env, best_nb);
    } else if (trees <= 48) {
        for (int rg = 1; rg <= rg_max && best_rg == 0; ++rg) { const int tt = tt_for(rg, 2); if (tt >= 4) { best_rg = rg; best_tt = std::min(tt, 8); } }
    } else {
        for (int rg = rg_max; rg >= 1 && best_rg == 0; --rg) { const int tt = tt_for(rg, 2); if (tt >= 8) { best_rg = rg; best_tt = tt; } }
        for (int rg = rg_max; rg >= 1 && best_rg == 0; --rg) { const int tt = tt_for(rg, 2); if (tt >= 4) { best_rg = rg; best_tt = tt; } }
    }
    if (best_rg == 0 || best_tt < 4) return false;
    if (tt_env >= 4 && tt_env <= best_tt) best_tt = tt_env & ~3;
    pl.RG = best_rg;
    pl.TT = best_tt;
    pl.NB = best_nb;
    pl.lds = lds_for(best_rg, best_tt, best_nb);
    return true;
}

template <int DMAX, int MAXD, bool CAT>
static bool launch_obl2(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
                        int stop_tree, float *out, hipStream_t s) {
    const int trees = pm.tree_chunk > 0 ? pm.tree_chunk : stop_tree - start_tree;
    Obl2Plan pl;
    if (!obl2_plan(F, Fc, CAT, MAXD, DMAX, trees, pl)) return false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static uint64_t attr_done = 0;   // per device
    if (dev < 64 && !((attr_done >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_predict_obl2<DMAX, MAXD, CAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done |= 1ull << dev;
    }
    Obl2Coef<DMAX> coef;
    for (int j = 0; j < DMAX; ++j) coef.lr[j] = j < pm.D ? pm.coef[j] : 0.0f;
    const int R = 64 * pl.RG;
    const int splits = pm.tree_chunk > 0 ? (stop_tree - start_tree + pm.tree_chunk - 1) / pm.tree_chunk : 1;
    hipLaunchKernelGGL((k_predict_obl2<DMAX, MAXD, CAT>), dim3((n + R - 1) / R, splits), dim3(256 * pl.RG), pl.lds, s, pm.values_sw, pm.cond_ra,
                       pm.bias, coef, pm.D, obs, F, cat_codes, Fc, n, start_tree, stop_tree, pm.tree_chunk > 0 ? pm.partial : out, R,
                       pl.TT, pl.NB, pl.xs, pm.tree_chunk);
    return true;
}

template <int DMAX, int MAXD>
static bool launch_obl2_c(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
                          int stop_tree, float *out, hipStream_t s) {
    return Fc > 0 ? launch_obl2<DMAX, MAXD, true>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)
                  : launch_obl2<DMAX, MAXD, false>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
}

template <int DMAX>
static bool launch_obl2_d(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
                          int stop_tree, float *out, hipStream_t s) {
    switch (pm.obl2_maxd) {
        case 4: return launch_obl2_c<DMAX, 4>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        case 6: return launch_obl2_c<DMAX, 6>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        case 8: return launch_obl2_c<DMAX, 8>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        default: return false;
    }
}

}  // namespace

int obl2_padded_outputs(int D) {
    if (D <= 4) return 4;
    if (D <= 8) return 8;
    if (D <= 16) return 16;
    if (D <= 32) return 32;
    return D <= 64 ? 64 : 0;
}
int obl2_levels(int max_depth) { return max_depth <= 4 ? 4 : max_depth <= 6 ? 6 : max_depth <= 8 ? 8 : 0; }

bool predict_obl2(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree, int stop_tree,
                  float *out, hipStream_t s) {
    if (pm.values_sw == nullptr || pm.cond_ra == nullptr || pm.obl2_maxd == 0) return false;
    if (Fc > 0 && (cat_codes == nullptr || pm.cat_dict_size > 65535)) return false;
    switch (obl2_padded_outputs(pm.D)) {
        case 4: return launch_obl2_d<4>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        case 8: return launch_obl2_d<8>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        case 16: return launch_obl2_d<16>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        case 32: return launch_obl2_d<32>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        case 64: return launch_obl2_d<64>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
        default: return false;
    }
}

}  // namespace kern
}  // namespace gbrl
