// predict.hip -- gfx950 kernels behind GBRL::predict (A13): the general kernels (any ensemble the file format can hold) and the
// fast paths for oblivious and greedy ensembles, plus the dispatcher kern::predict.
#include "kernels.h"
#include "hooks.h"
#include "kernels_common.h"

#include <algorithm>
#include <stdexcept>
#include <cmath>

namespace gbrl {
namespace kern {

namespace {

// ------------------------------------------------------------------------------------------------------------
// A13  prediction: one thread per row, trees in order (Q14), pred = fma(-lr, value, pred) per optimizer range
// (optimizer.cpp:110-118, contracted in the reference build).  Oblivious: leaf index from the tree's conditions
// (predictor.cpp:231-265).  Greedy: walk the leaves in order, including the reference's behaviour of walking past a
// leaf that can never match (Q7, predictor.cpp:188-229).
// ------------------------------------------------------------------------------------------------------------
template <int DMAX>
__global__ __launch_bounds__(256) void k_predict(PredictModel pm, const float *__restrict__ obs, int F,
                                                 const int32_t *__restrict__ cat_codes, int Fc, int n, int start_tree,
                                                 int stop_tree, float *__restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int D = pm.D, md = pm.max_depth;
    float p[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) p[j] = j < D ? 0.0f + pm.bias[j] : 0.0f;
    const float *x = obs + static_cast<size_t>(row) * F;
    const int32_t *xc = cat_codes ? cat_codes + static_cast<size_t>(row) * Fc : nullptr;
    auto test = [&](int c) -> bool {
        const int f = pm.feature_indices[c];
        return pm.is_numerics[c] ? (x[f] > pm.feature_values[c]) : (xc != nullptr && xc[f] == pm.cat_ids[c]);
    };
    auto apply = [&](const float *v) {
        for (int o = 0; o < pm.n_opts; ++o) {
            const float lr = pm.opt_lr[o];
            const int a = pm.opt_start[o], b = pm.opt_stop[o];
#pragma unroll
            for (int j = 0; j < DMAX; ++j)
                if (j >= a && j < b) p[j] = fmaf(-lr, v[j], p[j]);
        }
    };
    if (stop_tree > start_tree && pm.n_opts > 0) {
        if (pm.oblivious) {
            for (int t = start_tree; t < stop_tree; ++t) {
                const int depth = pm.depths[t], cond = t * md;
                int leaf = 0;
                for (int d = 0; d < depth; ++d) leaf |= (test(cond + d) ? 1 : 0) << (depth - 1 - d);
                apply(pm.values + static_cast<size_t>(pm.tree_indices[t] + leaf) * D);
            }
        } else {
            int t = start_tree;
            int leaf = pm.tree_indices[t];
            while (leaf < pm.n_leaves && t < stop_tree) {
                const int depth = pm.depths[leaf], cond = leaf * md;
                bool passed = false;
                for (int d = depth - 1; d >= 0; --d) {
                    passed = (test(cond + d) == (pm.inequality_directions[cond + d] != 0));
                    if (!passed) break;
                }
                if (passed) {
                    apply(pm.values + static_cast<size_t>(leaf) * D);
                    ++t;
                    if (t < stop_tree) leaf = pm.tree_indices[t];
                } else {
                    ++leaf;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DMAX; ++j)
        if (j < D) out[static_cast<size_t>(row) * D + j] = p[j];
}

// Prediction v2: each block stages its R rows of observations in LDS once (coalesced read of the row-major matrix; row
// stride F+1 words so that lane r reading feature f hits bank (r+f)%32: conflict-free for the wave-uniform f of an
// oblivious tree), then walks all trees over the tile.  Oblivious ensembles additionally stage, TT trees at a time, the
// conditions and the leaf values in LDS, so the inner loop touches LDS only.  Per-row accumulation order = tree order,
// pred = fma(-lr, value, pred) (optimizer.cpp:110-118).
// Coalesced staging of rows [r0, r0+rows) of a row-major f32 matrix into an LDS tile with row stride xs (odd: a wave that
// reads one feature of 64 consecutive rows touches 64 different banks).  8 x 16-byte loads in flight per thread: a block of
// the predict kernels is one wave per SIMD, nothing else hides the HBM latency.
__device__ __forceinline__ void stage_row_tile(const float *__restrict__ obs, int r0, int rows, int F, int xs, float *__restrict__ xt) {
    const int R = blockDim.x;
    const float *src = obs + static_cast<size_t>(r0) * F;
    if (F > 0 && (F & 3) == 0) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        const int F4 = F >> 2, tot4 = rows * F4;
        constexpr int UL = 8;
        for (int i0 = threadIdx.x; i0 < tot4; i0 += R * UL) {
            float4 v[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = i0 + u * R;
                v[u] = i < tot4 ? src4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = i0 + u * R;
                if (i < tot4) {
                    const int r = i / F4, f = (i - r * F4) << 2;
                    float *dst = xt + r * xs + f;
                    dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
                }
            }
        }
    } else {
        const int tot = rows * F;
        for (int i = threadIdx.x; i < tot; i += R) {
            const int r = i / F, f = i - r * F;
            xt[r * xs + f] = src[i];
        }
    }
}

constexpr int kPredMaxOpts = 4;   // optimisers kept in registers (more fall back to the direct kernel)
template <int DMAX>
__global__ __launch_bounds__(256) void k_predict_tiled(PredictModel pm, const float *__restrict__ obs, int F,
                                                       const int32_t *__restrict__ cat_codes, int Fc, int n, int start_tree,
                                                       int stop_tree, float *__restrict__ out, int TT) {
    extern __shared__ float ptile[];
    const int R = blockDim.x;
    const int xs = F + 1;
    const int D = pm.D, md = pm.max_depth;
    const int DS = D | 1;                                     // odd leaf stride: a wave's random leaves spread over all LDS banks
    const int vstride = (1 << md) * DS;
    float *xt = ptile;                                        // [R][F+1]
    float *vt = ptile + static_cast<size_t>(R) * xs;          // [TT][2^md][D]      (oblivious only)
    int *ct = reinterpret_cast<int *>(vt + static_cast<size_t>(TT) * vstride);   // [TT][1 + 2*md]: depth, then (feature | ~cat, threshold bits | cat id)
    const int cstride = 1 + 2 * md;
    const int r0 = blockIdx.x * R;
    const int rows = min(R, n - r0);
    stage_row_tile(obs, r0, rows, F, xs, xt);
    const int row = r0 + threadIdx.x;
    const bool live = threadIdx.x < rows;
    float p[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) p[j] = j < D ? 0.0f + pm.bias[j] : 0.0f;
    const float *x = xt + threadIdx.x * xs;
    const int32_t *xc = (cat_codes && live) ? cat_codes + static_cast<size_t>(row) * Fc : nullptr;
    float olr[kPredMaxOpts];
    int oa[kPredMaxOpts], ob[kPredMaxOpts];
#pragma unroll
    for (int o = 0; o < kPredMaxOpts; ++o) {
        const bool on = o < pm.n_opts;
        olr[o] = on ? pm.opt_lr[o] : 0.0f;
        oa[o] = on ? pm.opt_start[o] : 0;
        ob[o] = on ? pm.opt_stop[o] : 0;
    }
    auto apply = [&](const float *v) {
        float vv[DMAX];
#pragma unroll
        for (int j = 0; j < DMAX; ++j) vv[j] = j < D ? v[j] : 0.0f;
#pragma unroll
        for (int o = 0; o < kPredMaxOpts; ++o) {
#pragma unroll
            for (int j = 0; j < DMAX; ++j)
                if (j >= oa[o] && j < ob[o]) p[j] = fmaf(-olr[o], vv[j], p[j]);
        }
    };
    __syncthreads();
    if (stop_tree > start_tree && pm.n_opts > 0) {
        if (pm.oblivious) {
            for (int t0 = start_tree; t0 < stop_tree; t0 += TT) {
                const int tn = min(TT, stop_tree - t0);
                __syncthreads();   // previous tile fully consumed
                // staging of the group's trees, flattened over (tree, element) so that the loads of different trees are in
                // flight together (a per-tree loop costs one global-memory round trip per tree and block)
                {
                    const int vmax = (1 << md) * D;
                    constexpr int US = 4;
                    for (int i0 = threadIdx.x; i0 < tn * vmax; i0 += R * US) {
                        float v[US];
                        int dst[US];
#pragma unroll
                        for (int u = 0; u < US; ++u) {
                            const int i = i0 + u * R;
                            dst[u] = -1;
                            v[u] = 0.0f;
                            if (i < tn * vmax) {
                                const int tt = i / vmax, e = i - tt * vmax, t = t0 + tt;
                                if (e < (D << pm.depths[t])) {
                                    v[u] = pm.values[static_cast<size_t>(pm.tree_indices[t]) * D + e];
                                    dst[u] = tt * vstride + (e / D) * DS + (e % D);
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < US; ++u)
                            if (dst[u] >= 0) vt[dst[u]] = v[u];
                    }
                    for (int i = threadIdx.x; i < tn * (md + 1); i += R) {
                        const int tt = i / (md + 1), d = i - tt * (md + 1) - 1, t = t0 + tt;
                        const int depth = pm.depths[t];
                        if (d < 0) {
                            ct[tt * cstride] = depth;
                        } else if (d < depth) {
                            const int c = t * md + d;
                            const bool num = pm.is_numerics[c] != 0;
                            ct[tt * cstride + 1 + 2 * d] = num ? pm.feature_indices[c] : ~pm.feature_indices[c];
                            ct[tt * cstride + 2 + 2 * d] = num ? __float_as_int(pm.feature_values[c]) : pm.cat_ids[c];
                        }
                    }
                }
                __syncthreads();
                if (live) {
                    if (md <= 8) {
                        // All (feature, threshold) pairs of a tree are fetched before the first comparison and the row's
                        // features after them, two trees at a time: a thread's LDS round trips per tree drop from 2*depth
                        // dependent ones to ~2 (the block is one wave per SIMD, so nothing else hides that latency).
                        auto leaf_of = [&](const int *cc, int depth, const int (&fi)[8], const int (&tv)[8]) -> int {
                            (void)cc;
                            float xv[8];
#pragma unroll
                            for (int d = 0; d < 8; ++d) xv[d] = (d < depth && fi[d] >= 0) ? x[fi[d]] : 0.0f;
                            int leaf = 0;
#pragma unroll
                            for (int d = 0; d < 8; ++d) {
                                if (d < depth) {
                                    const bool pass = fi[d] >= 0 ? (xv[d] > __int_as_float(tv[d])) : (xc != nullptr && xc[~fi[d]] == tv[d]);
                                    leaf |= (pass ? 1 : 0) << (depth - 1 - d);
                                }
                            }
                            return leaf;
                        };
                        auto load_conds = [&](const int *cc, int depth, int (&fi)[8], int (&tv)[8]) {
#pragma unroll
                            for (int d = 0; d < 8; ++d) {
                                fi[d] = d < depth ? cc[1 + 2 * d] : 0;
                                tv[d] = d < depth ? cc[2 + 2 * d] : 0;
                            }
                        };
                        int tt = 0;
                        for (; tt + 1 < tn; tt += 2) {
                            const int *c0 = ct + tt * cstride, *c1 = c0 + cstride;
                            const int d0 = c0[0], d1 = c1[0];
                            int fi0[8], tv0[8], fi1[8], tv1[8];
                            load_conds(c0, d0, fi0, tv0);
                            load_conds(c1, d1, fi1, tv1);
                            const int l0 = leaf_of(c0, d0, fi0, tv0);
                            const int l1 = leaf_of(c1, d1, fi1, tv1);
                            apply(vt + tt * vstride + l0 * DS);          // per-row accumulation stays in tree order (Q14)
                            apply(vt + (tt + 1) * vstride + l1 * DS);
                        }
                        if (tt < tn) {
                            const int *c0 = ct + tt * cstride;
                            const int d0 = c0[0];
                            int fi0[8], tv0[8];
                            load_conds(c0, d0, fi0, tv0);
                            apply(vt + tt * vstride + leaf_of(c0, d0, fi0, tv0) * DS);
                        }
                    } else {
                        for (int tt = 0; tt < tn; ++tt) {
                            const int *cc = ct + tt * cstride;
                            const int depth = cc[0];
                            int leaf = 0;
                            for (int d = 0; d < depth; ++d) {
                                const int fi = cc[1 + 2 * d], tv = cc[2 + 2 * d];
                                const bool pass = fi >= 0 ? (x[fi] > __int_as_float(tv)) : (xc != nullptr && xc[~fi] == tv);
                                leaf |= (pass ? 1 : 0) << (depth - 1 - d);
                            }
                            apply(vt + tt * vstride + leaf * DS);
                        }
                    }
                }
            }
        } else if (live) {
            auto test = [&](int c) -> bool {
                const int f = pm.feature_indices[c];
                return pm.is_numerics[c] ? (x[f] > pm.feature_values[c]) : (xc != nullptr && xc[f] == pm.cat_ids[c]);
            };
            int t = start_tree;
            int leaf = pm.tree_indices[t];
            while (leaf < pm.n_leaves && t < stop_tree) {
                const int depth = pm.depths[leaf], cond = leaf * md;
                bool passed = false;
                for (int d = depth - 1; d >= 0; --d) {
                    passed = (test(cond + d) == (pm.inequality_directions[cond + d] != 0));
                    if (!passed) break;
                }
                if (passed) {
                    apply(pm.values + static_cast<size_t>(leaf) * D);
                    ++t;
                    if (t < stop_tree) leaf = pm.tree_indices[t];
                } else {
                    ++leaf;
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < DMAX; ++j)
            if (j < D) out[static_cast<size_t>(row) * D + j] = p[j];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// A13 fast path: oblivious, all-numeric ensembles with non-overlapping optimiser ranges (the common case).
// The traversal is VALU-issue bound (one wave per SIMD: a block's row tile fills the LDS), so the work per (row, tree) is cut
// to the minimum: the tree's (feature, threshold) pairs come through the SCALAR cache (uniform addresses -> s_load, no
// vector or LDS instruction), a level costs an address add, an LDS read of the row's feature, a compare and a select/or,
// the leaf's D values sit transposed in LDS ([tree][d][leaf]: one base address, immediate offsets) and the optimiser ranges
// are folded into one learning rate per output.  Accumulation order per row is still tree order (Q14).
// ------------------------------------------------------------------------------------------------------------
template <int DMAX>
struct OblCoef { float lr[DMAX]; };   // one learning rate per output (0 for the padded outputs j >= D)

template <int DMAX, int MAXD, bool CAT>
__global__ __launch_bounds__(256) void k_predict_obl(const float *__restrict__ values, const int32_t *__restrict__ tree_indices,
                                                     const int32_t *__restrict__ cond_pack, const int32_t *__restrict__ depths,
                                                     const float *__restrict__ bias, OblCoef<DMAX> coef, int D, int md,
                                                     const float *__restrict__ obs, int F, const int32_t *__restrict__ cat_codes, int Fc, int n,
                                                     int start_tree, int stop_tree, float *__restrict__ out, int TT, int tree_chunk) {
    extern __shared__ float ptile[];
    // tree_chunk > 0: this block covers the trees [start + y*chunk, ...) only and writes a PARTIAL sum (no bias) into slice y of `out`
    // (small batches with large ensembles: the trees are spread over blocks; k_predict_combine adds the slices in tree order)
    if (tree_chunk > 0) {
        start_tree += blockIdx.y * tree_chunk;
        if (blockIdx.y + 1 < gridDim.y) stop_tree = min(stop_tree, start_tree + tree_chunk);   // the last slice takes the remainder
        out += static_cast<size_t>(blockIdx.y) * n * D;
    }
    const int R = blockDim.x;
    const int xs = F | 1;
    const int LS = 1 << md;
    float *xt = ptile;                                        // [R][xs]
    float *vt = ptile + static_cast<size_t>(R) * xs;          // [TT][DMAX][LS]  (outputs j >= D are zero columns)
    int *tmeta = reinterpret_cast<int *>(vt + static_cast<size_t>(TT) * DMAX * LS);   // [TT][2]: first leaf, leaves of the tree
    const int r0 = blockIdx.x * R;
    const int rows = min(R, n - r0);
    stage_row_tile(obs, r0, rows, F, xs, xt);
    const bool live = static_cast<int>(threadIdx.x) < rows;
    float p[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) p[j] = (j < D && tree_chunk == 0) ? 0.0f + bias[j] : 0.0f;
    const float *x = xt + threadIdx.x * xs;
    // categorical conditions (feature word < 0) compare the row's dictionary id; the branch is uniform (scalar condition words)
    const int32_t *xc = (CAT && live) ? cat_codes + static_cast<size_t>(r0 + threadIdx.x) * Fc : nullptr;
    constexpr int kTreeElems = DMAX;   // per leaf
    const int vtree = kTreeElems * LS;
    for (int t0 = start_tree; t0 < stop_tree; t0 += TT) {
        const int tn = min(TT, stop_tree - t0);
        __syncthreads();   // row tile staged / previous group fully consumed
        // per-tree (first leaf, leaf count) of the group go to LDS first, so that the value loads below depend on nothing but
        // an LDS read and all of a thread's loads are in flight together (a dependent global load chain per element costs a
        // memory round trip per tree and block -- the whole kernel's time for small ensembles)
        int full_depth = 1;
        if (static_cast<int>(threadIdx.x) < tn) {
            const int t = t0 + threadIdx.x;
            const int dep = depths[t];
            tmeta[2 * threadIdx.x] = tree_indices[t];
            tmeta[2 * threadIdx.x + 1] = 1 << dep;
            full_depth = dep == MAXD;
        }
        const bool group_full = __syncthreads_and(full_depth) != 0;   // every tree of the group has MAXD levels (the usual case)
        {
            constexpr int US = 8;
            for (int i0 = threadIdx.x; i0 < tn * vtree; i0 += R * US) {
                float v[US];
                int dst[US];
#pragma unroll
                for (int u = 0; u < US; ++u) {
                    const int i = i0 + u * R;
                    dst[u] = -1;
                    v[u] = 0.0f;
                    if (i < tn * vtree) {
                        // i = (tt, leaf, j) with j fastest: consecutive threads read consecutive values of a leaf row
                        const int tt = i / vtree, e = i - tt * vtree;
                        const int leaf = e / DMAX, j = e - leaf * DMAX;
                        dst[u] = (tt * DMAX + j) * LS + leaf;
                        if (j < D && leaf < tmeta[2 * tt + 1]) v[u] = values[(static_cast<size_t>(tmeta[2 * tt]) + leaf) * D + j];
                    }
                }
#pragma unroll
                for (int u = 0; u < US; ++u)
                    if (dst[u] >= 0) vt[dst[u]] = v[u];
            }
        }
        __syncthreads();
        if (live) {
            auto apply = [&](int tt, int leaf) {   // DMAX reads at immediate offsets from one address, DMAX fused multiply-adds
                const float *v = vt + tt * vtree + leaf;
                float vv[DMAX];
#pragma unroll
                for (int j = 0; j < DMAX; ++j) vv[j] = v[j * LS];
#pragma unroll
                for (int j = 0; j < DMAX; ++j) p[j] = fmaf(-coef.lr[j], vv[j], p[j]);
            };
            // Full-depth groups run branch-free: the 2*MAXD condition words of a tree arrive in wide scalar loads, its MAXD
            // feature reads are issued back to back, and four trees are in flight at once (independent chains, so their scalar /
            // LDS latencies overlap -- a block is one wave per SIMD, nothing else hides them).  Leaf values are applied strictly
            // in tree order.
            auto leaf_full = [&](int t) -> int {
                const int32_t *cp = cond_pack + static_cast<size_t>(t) * 2 * MAXD;
                int fi[MAXD];
                float tv[MAXD], xv[MAXD];
#pragma unroll
                for (int d = 0; d < MAXD; ++d) { fi[d] = cp[2 * d]; tv[d] = __int_as_float(cp[2 * d + 1]); }
#pragma unroll
                for (int d = 0; d < MAXD; ++d) xv[d] = (!CAT || fi[d] >= 0) ? x[fi[d]] : 0.0f;
                int leaf = 0;
#pragma unroll
                for (int d = 0; d < MAXD; ++d) {
                    const bool pass = (!CAT || fi[d] >= 0) ? (xv[d] > tv[d]) : (xc[~fi[d]] == __float_as_int(tv[d]));
                    leaf |= pass ? (1 << (MAXD - 1 - d)) : 0;
                }
                return leaf;
            };
            auto leaf_any = [&](int t) -> int {    // shallower trees (growth stopped early) or max_depth < MAXD
                const int depth = depths[t];
                const int32_t *cp = cond_pack + static_cast<size_t>(t) * 2 * md;
                int leaf = 0;
                for (int d = 0; d < depth; ++d) {
                    const int fi = cp[2 * d], tv = cp[2 * d + 1];
                    const bool pass = (!CAT || fi >= 0) ? (x[fi] > __int_as_float(tv)) : (xc[~fi] == tv);
                    leaf |= pass ? (1 << (depth - 1 - d)) : 0;
                }
                return leaf;
            };
            int tt = 0;
            if (group_full && md == MAXD) {
                for (; tt + 3 < tn; tt += 4) {
                    const int l0 = leaf_full(t0 + tt), l1 = leaf_full(t0 + tt + 1), l2 = leaf_full(t0 + tt + 2), l3 = leaf_full(t0 + tt + 3);
                    apply(tt, l0); apply(tt + 1, l1); apply(tt + 2, l2); apply(tt + 3, l3);
                }
                for (; tt < tn; ++tt) apply(tt, leaf_full(t0 + tt));
            } else {
                for (; tt < tn; ++tt) apply(tt, leaf_any(t0 + tt));
            }
        }
    }
    if (live) {
        float *o = out + static_cast<size_t>(r0 + threadIdx.x) * D;
#pragma unroll
        for (int j = 0; j < DMAX; ++j)
            if (j < D) o[j] = p[j];
    }
}

template <int DMAX, int MAXD, bool CAT>
static bool launch_predict_obl(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
                               int stop_tree, float *out, hipStream_t s) {
    const size_t budget = 156 * 1024;   // leaves room for the kernel's small static LDS (block-wide vote)
    const size_t vtree = (static_cast<size_t>(1) << pm.max_depth) * DMAX * sizeof(float);
    const int xs = F | 1;
    const size_t meta = 2 * 64 * sizeof(int);
    int R = 256;
    while (R >= 64 && static_cast<size_t>(R) * xs * sizeof(float) + vtree + meta > budget) R -= 64;
    if (R < 64) return false;
    int TT = static_cast<int>((budget - meta - static_cast<size_t>(R) * xs * sizeof(float)) / vtree);
    TT = std::max(1, std::min(TT, 64));
    const size_t lds = static_cast<size_t>(R) * xs * sizeof(float) + static_cast<size_t>(TT) * vtree + meta;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_predict_obl<DMAX, MAXD, CAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    }
    OblCoef<DMAX> coef;
    for (int j = 0; j < DMAX; ++j) coef.lr[j] = j < pm.D ? pm.coef[j] : 0.0f;
    const int splits = pm.tree_chunk > 0 ? pm.tree_splits : 1;
    hipLaunchKernelGGL((k_predict_obl<DMAX, MAXD, CAT>), dim3((n + R - 1) / R, splits), dim3(R), lds, s, pm.values, pm.tree_indices, pm.cond_pack,
                       pm.depths, pm.bias, coef, pm.D, pm.max_depth, obs, F, cat_codes, Fc, n, start_tree, stop_tree,
                       pm.tree_chunk > 0 ? pm.partial : out, TT, pm.tree_chunk);
    return true;
}
template <int DMAX, bool CAT>
static bool launch_predict_obl_c(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
                                 int stop_tree, float *out, hipStream_t s) {
    if (pm.max_depth <= 4) return launch_predict_obl<DMAX, 4, CAT>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
    if (pm.max_depth <= 6) return launch_predict_obl<DMAX, 6, CAT>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
    if (pm.max_depth <= 8) return launch_predict_obl<DMAX, 8, CAT>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
    return false;
}
template <int DMAX>
static bool launch_predict_obl_d(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
                                 int stop_tree, float *out, hipStream_t s) {
    return Fc > 0 ? launch_predict_obl_c<DMAX, true>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)
                  : launch_predict_obl_c<DMAX, false>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s);
}

// ------------------------------------------------------------------------------------------------------------
// A13 fast path for GREEDY ensembles.  The reference finds a row's leaf by testing the leaves of a tree one after the other
// until all conditions of one pass (predictor.cpp:188-229): up to leaves x depth tests.  Leaves of a proper tree partition
// the space, so the first leaf that passes is THE leaf the row reaches by descending the tree; the host rebuilds that tree from
// the leaves' paths (Engine::sync_model_to_device) and the kernel descends it: <= depth steps per (row, tree).  Ensembles that
// are not proper trees, or contain a depth-0 tree (whose leaf never passes and lets the reference's walk run on, Q7), keep the
// generic kernel.  Same row tile / transposed leaf values / per-output learning rates / tree-order accumulation as
// k_predict_obl; a group's nodes are staged in LDS next to its values.
// ------------------------------------------------------------------------------------------------------------
template <int DMAX>
__global__ __launch_bounds__(256) void k_predict_grd(const float *__restrict__ values, const int32_t *__restrict__ tree_indices,
                                                     const int32_t *__restrict__ nodes, const int32_t *__restrict__ node_off,
                                                     const float *__restrict__ bias, OblCoef<DMAX> coef, int D, int n_leaves_total,
                                                     int n_trees_total, int max_nodes, int max_leaves,
                                                     const float *__restrict__ obs, int F, const int32_t *__restrict__ cat_codes, int Fc,
                                                     int n, int start_tree, int stop_tree, float *__restrict__ out, int TT, int tree_chunk) {
    extern __shared__ float ptile[];
    // tree_chunk > 0: this block covers the trees [start + y*chunk, ...) only and writes a PARTIAL sum (no bias) into slice y of `out`
    // (small batches with large ensembles: the trees are spread over blocks; k_predict_combine adds the slices in tree order)
    if (tree_chunk > 0) {
        start_tree += blockIdx.y * tree_chunk;
        if (blockIdx.y + 1 < gridDim.y) stop_tree = min(stop_tree, start_tree + tree_chunk);   // the last slice takes the remainder
        out += static_cast<size_t>(blockIdx.y) * n * D;
    }
    const int R = blockDim.x;
    const int xs = F | 1;
    const int LS = max_leaves;
    float *xt = ptile;                                              // [R][xs]
    float *vt = ptile + static_cast<size_t>(R) * xs;                // [TT][DMAX][LS]
    int4 *nt = reinterpret_cast<int4 *>(vt + static_cast<size_t>(TT) * DMAX * LS);   // [TT][max_nodes]
    int *tmeta = reinterpret_cast<int *>(nt + static_cast<size_t>(TT) * max_nodes);  // [TT][4]: first leaf, leaves, first node, nodes
    const int r0 = blockIdx.x * R;
    const int rows = min(R, n - r0);
    stage_row_tile(obs, r0, rows, F, xs, xt);
    const bool live = static_cast<int>(threadIdx.x) < rows;
    float p[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) p[j] = (j < D && tree_chunk == 0) ? 0.0f + bias[j] : 0.0f;
    const float *x = xt + threadIdx.x * xs;
    const int32_t *xc = (cat_codes && live) ? cat_codes + static_cast<size_t>(r0 + threadIdx.x) * Fc : nullptr;
    const int vtree = DMAX * LS;
    for (int t0 = start_tree; t0 < stop_tree; t0 += TT) {
        const int tn = min(TT, stop_tree - t0);
        __syncthreads();
        if (static_cast<int>(threadIdx.x) < tn) {
            const int t = t0 + threadIdx.x;
            const int l0 = tree_indices[t], l1 = t + 1 < n_trees_total ? tree_indices[t + 1] : n_leaves_total;
            tmeta[4 * threadIdx.x + 0] = l0;
            tmeta[4 * threadIdx.x + 1] = l1 - l0;
            tmeta[4 * threadIdx.x + 2] = node_off[t];
            tmeta[4 * threadIdx.x + 3] = node_off[t + 1] - node_off[t];
        }
        __syncthreads();
        {
            constexpr int US = 8;
            for (int i0 = threadIdx.x; i0 < tn * vtree; i0 += R * US) {
                float v[US];
                int dst[US];
#pragma unroll
                for (int u = 0; u < US; ++u) {
                    const int i = i0 + u * R;
                    dst[u] = -1;
                    v[u] = 0.0f;
                    if (i < tn * vtree) {
                        const int tt = i / vtree, e = i - tt * vtree;
                        const int leaf = e / DMAX, j = e - leaf * DMAX;
                        dst[u] = (tt * DMAX + j) * LS + leaf;
                        if (j < D && leaf < tmeta[4 * tt + 1]) v[u] = values[(static_cast<size_t>(tmeta[4 * tt]) + leaf) * D + j];
                    }
                }
#pragma unroll
                for (int u = 0; u < US; ++u)
                    if (dst[u] >= 0) vt[dst[u]] = v[u];
            }
            const int4 *gn = reinterpret_cast<const int4 *>(nodes);
            for (int i = threadIdx.x; i < tn * max_nodes; i += R) {
                const int tt = i / max_nodes, k = i - tt * max_nodes;
                if (k < tmeta[4 * tt + 3]) nt[i] = gn[tmeta[4 * tt + 2] + k];
            }
        }
        __syncthreads();
        if (live) {
            auto leaf_of = [&](int tt) -> int {
                const int4 *tn_ = nt + tt * max_nodes;
                int node = tmeta[4 * tt + 3] > 0 ? 0 : -1;   // a single-leaf tree has no node (never reaches here: grd_ok excludes it)
                while (node >= 0) {
                    const int4 nd = tn_[node];
                    const bool right = nd.x >= 0 ? (x[nd.x] > __int_as_float(nd.y)) : (xc != nullptr && xc[~nd.x] == nd.y);
                    node = right ? nd.w : nd.z;
                }
                return ~node;
            };
            auto apply = [&](int tt, int leaf) {
                const float *v = vt + tt * vtree + leaf;
                float vv[DMAX];
#pragma unroll
                for (int j = 0; j < DMAX; ++j) vv[j] = v[j * LS];
#pragma unroll
                for (int j = 0; j < DMAX; ++j) p[j] = fmaf(-coef.lr[j], vv[j], p[j]);
            };
            int tt = 0;
            for (; tt + 3 < tn; tt += 4) {   // four descents in flight (independent chains); values applied in tree order
                int nd0 = 0, nd1 = 0, nd2 = 0, nd3 = 0;
                const int4 *b0 = nt + tt * max_nodes, *b1 = b0 + max_nodes, *b2 = b1 + max_nodes, *b3 = b2 + max_nodes;
                auto step = [&](const int4 *base, int &node) {
                    if (node >= 0) {
                        const int4 nd = base[node];
                        const bool right = nd.x >= 0 ? (x[nd.x] > __int_as_float(nd.y)) : (xc != nullptr && xc[~nd.x] == nd.y);
                        node = right ? nd.w : nd.z;
                    }
                };
                while ((nd0 & nd1 & nd2 & nd3) >= 0) {   // until all four are leaves (negative)
                    step(b0, nd0); step(b1, nd1); step(b2, nd2); step(b3, nd3);
                }
                apply(tt, ~nd0); apply(tt + 1, ~nd1); apply(tt + 2, ~nd2); apply(tt + 3, ~nd3);
            }
            for (; tt < tn; ++tt) apply(tt, leaf_of(tt));
        }
    }
    if (live) {
        float *o = out + static_cast<size_t>(r0 + threadIdx.x) * D;
#pragma unroll
        for (int j = 0; j < DMAX; ++j)
            if (j < D) o[j] = p[j];
    }
}

template <int DMAX>
static bool launch_predict_grd(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n,
                               int start_tree, int stop_tree, float *out, hipStream_t s) {
    const size_t budget = 156 * 1024;
    const int LS = pm.grd_max_leaves, MN = std::max(1, pm.grd_max_nodes);
    const size_t per_tree = static_cast<size_t>(LS) * DMAX * sizeof(float) + static_cast<size_t>(MN) * 16 + 16;
    const int xs = F | 1;
    int R = 256;
    while (R >= 64 && static_cast<size_t>(R) * xs * sizeof(float) + per_tree > budget) R -= 64;
    if (R < 64) return false;
    int TT = static_cast<int>((budget - static_cast<size_t>(R) * xs * sizeof(float)) / per_tree);
    TT = std::max(1, std::min(TT, 64));
    const size_t lds = static_cast<size_t>(R) * xs * sizeof(float) + static_cast<size_t>(TT) * per_tree;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_predict_grd<DMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    }
    OblCoef<DMAX> coef;
    for (int j = 0; j < DMAX; ++j) coef.lr[j] = j < pm.D ? pm.coef[j] : 0.0f;
    const int splits = pm.tree_chunk > 0 ? pm.tree_splits : 1;
    hipLaunchKernelGGL((k_predict_grd<DMAX>), dim3((n + R - 1) / R, splits), dim3(R), lds, s, pm.values, pm.tree_indices, pm.grd_nodes,
                       pm.grd_node_off, pm.bias, coef, pm.D, pm.n_leaves, pm.n_trees, MN, LS, obs, F, cat_codes, Fc, n, start_tree,
                       stop_tree, pm.tree_chunk > 0 ? pm.partial : out, TT, pm.tree_chunk);
    return true;
}

template <int DMAX>
static bool launch_predict_tiled(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n,
                                 int start_tree, int stop_tree, float *out, hipStream_t s) {
    if (F <= 0 || pm.n_opts > kPredMaxOpts) return false;
    if (pm.max_depth > 256) return false;
    const size_t budget = 159 * 1024;
    const size_t vtree = pm.oblivious ? ((static_cast<size_t>(1) << pm.max_depth) * (pm.D | 1) + 1 + 2 * pm.max_depth) * sizeof(float) : 0;
    int R = 256;
    while (R >= 64 && static_cast<size_t>(R) * (F + 1) * sizeof(float) + vtree > budget) R -= 64;
    if (R < 64) return false;   // rows too wide for an LDS tile: the caller uses the direct kernel
    int TT = 1;
    if (pm.oblivious) {
        TT = static_cast<int>((budget - static_cast<size_t>(R) * (F + 1) * sizeof(float)) / vtree);
        if (TT > 64) TT = 64;
        if (TT < 1) return false;
    }
    const size_t lds = static_cast<size_t>(R) * (F + 1) * sizeof(float) + static_cast<size_t>(pm.oblivious ? TT : 0) * vtree;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_predict_tiled<DMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(k_predict_tiled<DMAX>, dim3((n + R - 1) / R), dim3(R), lds, s, pm, obs, F, cat_codes, Fc, n, start_tree,
                       stop_tree, out, TT);
    return true;
}

// out[r][j] = bias[j] + partial[0][r][j] + partial[1][r][j] + ...  (slices in tree order)
__global__ void k_predict_combine(const float *__restrict__ partial, int splits, size_t n_el, int D, const float *__restrict__ bias,
                                  float *__restrict__ out) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_el) return;
    float p = 0.0f + bias[i % D];
    for (int q = 0; q < splits; ++q) p += partial[static_cast<size_t>(q) * n_el + i];
    out[i] = p;
}

void predict(const PredictModel &pm_in, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree,
             int stop_tree, float *out, hipStream_t s) {
    // Small batches with large ensembles (an RL agent acting: tens to thousands of rows, hundreds to thousands of trees) would walk
    // every tree inside a handful of blocks.  The fast kernels then spread the trees over blocks (partial sums per tree range,
    // combined in tree order).  The float sums are associated differently from the one-chain-per-row order of large batches (the
    // reference's CPU path does the same for small batches: per-thread partial buffers, predictor.cpp:144-184); both are within
    // the 1e-5 parity tolerance, and large batches keep the exact tree-order chain (Q14).
    PredictModel pm = pm_in;
    pm.tree_chunk = 0;
    const int trees = stop_tree - start_tree;
    const int row_tiles = (n + 255) / 256;
    // Window 128..2048 trees: the float32 chain of the reference itself drifts from the exact sum by about 2e-9 per tree (9e-5 at
    // 50 000 trees, measured), so beyond a few thousand trees a differently associated -- more accurate -- sum would leave the 1e-5
    // band around the reference's chain; larger ensembles keep the chain wherever the reference runs it.
    pm.tree_splits = 1;
    // (up to 1024 rows the exact chain of kern::predict_chain costs the same -- 17 vs 14 us at 600 trees, 36 vs 32 us at 2000 -- and is taken
    // instead: those batches get the same bits whatever their size)
    const bool chain_first = pm.slots != nullptr && n <= 1024;
    if (pm.partial && trees >= 128 && trees <= 2048 && row_tiles <= 64 && !chain_first) {
        int splits = std::min(64, std::min(trees / 32, std::max(1, 512 / row_tiles)));
        while (splits > 1 && static_cast<size_t>(splits) * n * pm.D > pm.partial_floats) --splits;
        if (splits > 1) {
            pm.tree_chunk = (trees + splits - 1) / splits;
            pm.tree_splits = (trees + pm.tree_chunk - 1) / pm.tree_chunk;
        }
    } else if (pm.partial && trees > 2048) {
        // Larger ensembles, fewer than 2 * par_th rows (an agent acting in a few environments): the reference does NOT run the chain
        // there on any multi-threaded host.  predict_cpu (predictor.cpp:142-163) gives every OpenMP thread trees/n_tree_threads
        // consecutive trees (the last thread the remainder) and adds the per-thread buffers to bias in thread order whenever
        // n_tree_threads > n_sample_threads, with n_x_threads = clamp(count / par_th, 1, host threads) (utils.h:64-80) -- and
        // rows / par_th <= 1 makes n_sample_threads 1.  The same slices, for a nominal host of 64 threads: the result is the one the
        // reference produces on such a host (bit for bit, tests/test_gpu_edges.py), and the trees spread over 64 block columns
        // instead of one.  With more rows the reference's choice depends on the host (8 threads: the chain from 80 rows on), so
        // those batches keep the exact chain (kern::predict_chain below).
        const int par = pm.par_th > 0 ? pm.par_th : 1;
        const int n_tree_thr = std::max(1, std::min(64, trees / par));
        if (n / par <= 1 && n_tree_thr > 1 && static_cast<size_t>(n_tree_thr) * n * pm.D <= pm.partial_floats) {
            pm.tree_splits = n_tree_thr;
            pm.tree_chunk = trees / n_tree_thr;
        }
    }
    if (pm.tree_chunk == 0 && pm.slots && predict_chain(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) return;
    struct Combine {   // runs after whichever fast kernel took the launch
        const PredictModel &pm; int n, trees; float *out; hipStream_t s;
        void operator()() const {
            const int splits = pm.tree_splits;
            const size_t n_el = static_cast<size_t>(n) * pm.D;
            hipLaunchKernelGGL(k_predict_combine, dim3(static_cast<unsigned>((n_el + 255) / 256)), dim3(256), 0, s, pm.partial, splits, n_el, pm.D,
                               pm.bias, out);
        }
    } combine{pm, n, trees, out, s};
    // fast path: oblivious, every output updated by exactly one optimiser.  Outputs are padded to the template's DMAX (zero columns, zero
    // learning rates), so the steps between 8 and 32 are fine-grained: the work per (row, tree) grows with DMAX, not with D
    const uint64_t all_out = pm.D >= 64 ? ~0ull : ((1ull << pm.D) - 1ull);
    if (pm.oblivious && pm.obl_ok && pm.coef_ok && pm.coef_cover == all_out && (F > 0 || Fc > 0) && pm.D <= 64 && stop_tree > start_tree) {
        if (pm.tree_chunk == 0 && pm.obl2_maxd != 0 && predict_reg(pm, obs, F, Fc, n, start_tree, stop_tree, out, s)) return;
        if (pm.tree_chunk == 0 && pm.obl2_maxd != 0 && predict_pc(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) return;
        if (const char *e = hooks::raw(hooks::PREDICT_REG_ONLY))   // test hook: a shape the tests expect the register-tile kernel to take
            if (e[0] == '1') throw std::runtime_error("GBRL_HIP_PREDICT_REG_ONLY=1: the register-tile kernel does not take this shape");
        if (predict_obl2(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; }
        if (pm.D <= 4) { if (launch_predict_obl_d<4>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 8) { if (launch_predict_obl_d<8>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 12) { if (launch_predict_obl_d<12>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 16) { if (launch_predict_obl_d<16>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 20) { if (launch_predict_obl_d<20>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 24) { if (launch_predict_obl_d<24>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 32) { if (launch_predict_obl_d<32>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 48) { if (launch_predict_obl_d<48>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else { if (launch_predict_obl_d<64>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
    }
    // fast path: greedy ensembles whose trees were rebuilt as binary trees (descent instead of the leaf-by-leaf walk)
    if (!pm.oblivious && pm.grd_ok && pm.coef_ok && pm.coef_cover == all_out && (F > 0 || Fc > 0) && pm.D <= 64 && stop_tree > start_tree &&
        pm.grd_max_leaves <= 256) {
        if (pm.obl2_maxd != 0 && predict_grd_stream(pm, obs, F, Fc, n, start_tree, stop_tree, out, s)) return;
        if (predict_obl2(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; }
        if (pm.D <= 4) { if (launch_predict_grd<4>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 8) { if (launch_predict_grd<8>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 12) { if (launch_predict_grd<12>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 16) { if (launch_predict_grd<16>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 20) { if (launch_predict_grd<20>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 24) { if (launch_predict_grd<24>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 32) { if (launch_predict_grd<32>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else if (pm.D <= 48) { if (launch_predict_grd<48>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
        else { if (launch_predict_grd<64>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) { if (pm.tree_chunk > 0) combine(); return; } }
    }
    pm.tree_chunk = 0;   // the general kernels below always take the whole range
    if (pm.D <= 8) { if (launch_predict_tiled<8>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) return; }
    else if (pm.D <= 32) { if (launch_predict_tiled<32>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) return; }
    else { if (launch_predict_tiled<128>(pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out, s)) return; }
    dim3 grid((n + 255) / 256), block(256);
    if (pm.D <= 8)
        hipLaunchKernelGGL(k_predict<8>, grid, block, 0, s, pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out);
    else if (pm.D <= 32)
        hipLaunchKernelGGL(k_predict<32>, grid, block, 0, s, pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out);
    else
        hipLaunchKernelGGL(k_predict<128>, grid, block, 0, s, pm, obs, F, cat_codes, Fc, n, start_tree, stop_tree, out);
}

}  // namespace kern
}  // namespace gbrl
