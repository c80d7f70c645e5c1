// predict_chain.hip -- GBRL::predict (A13) for small and medium batches against large ensembles, in two launches.
//
// The tiled kernels (predict.hip, predict_obl2.hip) give every block 64..256 rows and walk the whole tree range inside the block:
// right for 2^20 rows, but with a few thousand rows only a few dozen blocks exist and each of them pays one LDS staging round trip
// per group of trees -- about 80 ns per tree whatever the batch (1.6 ms for 20 000 trees, measured).  The sum itself needs far
// less: per (row, output) it is ONE chain of fused multiply-adds in tree order, pred = fma(-lr, value, pred) (optimizer.cpp:110-118,
// Q14), and finding a row's leaf in a tree is independent of every other tree.  So:
//
//   k_leaf_slots   order-free, one thread per tree x a tile of rows staged in LDS: the offset of the leaf's values row,
//                  slots[row][t] = byte offset of that row = (first leaf of tree t + leaf the row falls into) * D * 4.  n * trees independent searches spread over
//                  the whole chip (oblivious: the packed conditions of predict.hip; greedy: its node records).
//   k_chain_relay  the chains in tree order, nothing else: the value gathers do not depend on the chain, so the waves of a block
//                  take turns on the same chains and gather while the others apply (see the kernel).
//
// The result has the bits of the one-chain-per-row kernels (same operands, same order, same fused operation), so it also serves
// fit()'s internal predictions.  Scratch: n * (trees + 704) int32.
#include "kernels.h"
#include "kernels_common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace gbrl {
namespace kern {

namespace {

constexpr int kChainU = 64;          // slot rows are padded in multiples of this many trees
constexpr int kSlotPad = 10 * kChainU;   // the relay requests up to 3 W - 2 batches behind the range (k_chain_relay asserts it)

struct ChainCoef {
    float c[64];        // -lr of the optimiser that owns the output
    uint64_t cover;     // outputs owned by an optimiser (the others keep the bias, like the general kernel)
};

template <bool GREEDY, int LV>
__global__ __launch_bounds__(256) void k_leaf_slots(const int32_t *__restrict__ tree_indices, const int32_t *__restrict__ depths,
                                                    const int32_t *__restrict__ cond_pack, int md, const int32_t *__restrict__ nodes,
                                                    const int32_t *__restrict__ node_off, int D, const float *__restrict__ obs, int F,
                                                    const int32_t *__restrict__ cat_codes, int Fc, int n, int start_tree, int Tn, int Ts,
                                                    int rows_per_block, int32_t *__restrict__ slots) {
    extern __shared__ float xt[];   // [rows_per_block][xs]
    const int xs = F | 1;
    const int r0 = blockIdx.y * rows_per_block;
    const int rows = min(rows_per_block, n - r0);
    const float *src = obs + static_cast<size_t>(r0) * F;
    if ((F & 3) == 0) {   // 16-byte reads, four in flight per thread: a block pays one memory round trip for its tile, not one per element
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        const int F4 = F >> 2, tot4 = rows * F4;
        for (int i0 = threadIdx.x; i0 < tot4; i0 += 256 * 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; v[u] = i < tot4 ? src4[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                if (i < tot4) {
                    const int r = i / F4, f = (i - r * F4) << 2;
                    float *dst4 = xt + r * xs + f;
                    dst4[0] = v[u].x; dst4[1] = v[u].y; dst4[2] = v[u].z; dst4[3] = v[u].w;
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < rows * F; i += 256) {
            const int r = i / F, f = i - r * F;
            xt[r * xs + f] = src[i];
        }
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;     // tree of the range, or padding behind it
    if (j >= Ts) return;
    int32_t *dst = slots + static_cast<size_t>(r0) * Ts + j;
    if (j >= Tn) {   // padding: the chain kernel loads (never applies) up to ten batches behind the range
        for (int r = 0; r < rows; ++r) dst[static_cast<size_t>(r) * Ts] = 0;
        return;
    }
    const int t = start_tree + j;
    const int first = tree_indices[t];
    if (!GREEDY) {
        const int depth = depths[t];
        const int32_t *cp = cond_pack + static_cast<size_t>(t) * 2 * md;
        if (depth <= LV) {
            // the tree's conditions in registers for all rows of the tile, padded at the FRONT with never-true ones (feature 0 >
            // +inf) so that level k always carries bit LV - 1 - k of an LV-level index whose top LV - depth bits are zero
            int fi[LV], tv[LV];
#pragma unroll
            for (int k = 0; k < LV; ++k) {
                const int dd = k - (LV - depth);
                fi[k] = dd >= 0 ? cp[2 * dd] : 0;
                tv[k] = dd >= 0 ? cp[2 * dd + 1] : 0x7f800000;
            }
            for (int r = 0; r < rows; ++r) {
                const float *x = xt + r * xs;
                int leaf = 0;
#pragma unroll
                for (int k = 0; k < LV; ++k) {
                    const bool pass = fi[k] >= 0 ? (x[fi[k]] > __int_as_float(tv[k]))
                                                 : (cat_codes != nullptr && cat_codes[static_cast<size_t>(r0 + r) * Fc + ~fi[k]] == tv[k]);
                    leaf |= pass ? ((1 << (LV - 1)) >> k) : 0;
                }
                dst[static_cast<size_t>(r) * Ts] = (first + leaf) * D * 4;
            }
        } else {
            for (int r = 0; r < rows; ++r) {
                const float *x = xt + r * xs;
                int leaf = 0;
                for (int d = 0; d < depth; ++d) {
                    const int fi = cp[2 * d], tv = cp[2 * d + 1];
                    const bool pass = fi >= 0 ? (x[fi] > __int_as_float(tv)) : (cat_codes != nullptr && cat_codes[static_cast<size_t>(r0 + r) * Fc + ~fi] == tv);
                    leaf |= pass ? (1 << (depth - 1 - d)) : 0;
                }
                dst[static_cast<size_t>(r) * Ts] = (first + leaf) * D * 4;
            }
        }
    } else {
        const int4 *tn = reinterpret_cast<const int4 *>(nodes) + node_off[t];
        const int n_nodes = node_off[t + 1] - node_off[t];
        for (int r = 0; r < rows; ++r) {
            const float *x = xt + r * xs;
            int node = n_nodes > 0 ? 0 : -1;
            while (node >= 0) {
                const int4 nd = tn[node];
                const bool right = nd.x >= 0 ? (x[nd.x] > __int_as_float(nd.y))
                                             : (cat_codes != nullptr && cat_codes[static_cast<size_t>(r0 + r) * Fc + ~nd.x] == nd.y);
                node = right ? nd.w : nd.z;
            }
            dst[static_cast<size_t>(r) * Ts] = (first + ~node) * D * 4;
        }
    }
}

// k_chain_relay: the chain, relayed between the W waves of a block.
// One wave alone cannot run it fast: a gather that misses L1 takes ~1500 clocks and a wave keeps at most ~64 loads in flight (the
// memory counter has six bits) -- one tree per 24..28 clocks per wave however the loads are arranged (measured with 2..9-stage
// register rings); more independent waves per CU do not help either, a CU retires one scattered 64-lane gather per ~15 clocks
// (8 waves per CU: 3.4 x slower than one), and 16-byte gathers for four adjacent outputs cost twice the 4-byte ones.  What helps is
// W waves that own the SAME 64 chains and take turns: wave w gathers the values of the batches w, w + W, w + 2W, ... (U trees each)
// into its registers, and when the running sums arrive in LDS it applies its U multiply-adds and hands them on.  Between two turns a
// wave has W - 1 turns of the others for its gathers and its next slot words.  G = outputs per lane (1: see above).
template <int W, int U, int G>
__global__ __launch_bounds__(64 * W) void k_chain_relay(const int32_t *__restrict__ slots, int Tn, int Ts, const float *__restrict__ values,
                                                        const float *__restrict__ bias, ChainCoef coef, int D, int n_lanes,
                                                        float *__restrict__ out) {
    static_assert((3 * W - 2) * U <= kSlotPad, "slot rows are padded for the batches requested behind the range");
    static_assert(G == 1 || G == 4, "one output or four adjacent outputs per lane");
    __shared__ float token[64 * G];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * 64 + lane;            // lane i = (row, group of G outputs)
    const bool live = i < n_lanes;
    const int ii = live ? i : n_lanes - 1;
    const int DG = D / G;
    const int row = ii / DG, d0 = (ii - row * DG) * G;
    float b0[G], c[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { b0[g] = 0.0f + bias[d0 + g]; c[g] = coef.c[d0 + g]; }
    const int32_t *sp = slots + static_cast<size_t>(row) * Ts;
    const char *vb8 = reinterpret_cast<const char *>(values);
    const uint32_t d4 = static_cast<uint32_t>(d0) * 4u;
    int s[U];
    float v[U][G];
    auto load_slots = [&](int batch) {
        const int4 *q = reinterpret_cast<const int4 *>(sp + static_cast<size_t>(batch) * U);
#pragma unroll
        for (int u = 0; u < U / 4; ++u) {
            const int4 x = q[u];
            s[4 * u] = x.x; s[4 * u + 1] = x.y; s[4 * u + 2] = x.z; s[4 * u + 3] = x.w;
        }
    };
    auto load_values = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const char *a = vb8 + (static_cast<uint32_t>(s[u]) + d4);
            if constexpr (G == 4) {
                const float4 x = *reinterpret_cast<const float4 *>(a);
                v[u][0] = x.x; v[u][1] = x.y; v[u][2] = x.z; v[u][3] = x.w;
            } else {
                v[u][0] = *reinterpret_cast<const float *>(a);
            }
        }
    };
    if (w == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) token[lane * G + g] = b0[g];
    }
    const int nbt = (Tn + U - 1) / U;                 // batches, the last one possibly partial
    const int rounds = (nbt + W - 1) / W;
    if (live) {
        load_slots(w);
        load_values();
        load_slots(w + W);
    }
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            __syncthreads();
            if (w == j) {                              // my turn: batch r * W + j
                const int kb = r * W + j;
                if (kb < nbt && live) {
                    float p[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) p[g] = token[lane * G + g];
                    if ((kb + 1) * U <= Tn) {
#pragma unroll
                        for (int u = 0; u < U; ++u)
#pragma unroll
                            for (int g = 0; g < G; ++g) p[g] = fmaf(c[g], v[u][g], p[g]);
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            if (kb * U + u < Tn) {
#pragma unroll
                                for (int g = 0; g < G; ++g) p[g] = fmaf(c[g], v[u][g], p[g]);
                            }
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) token[lane * G + g] = p[g];
                }
            } else if (w == (j + W - 1) % W && (r > 0 || j > 0)) {
                // my turn was the previous one: request my next batch (its slot words are here) and the slot words of the one after,
                // while the next wave applies its batch
                const int kb = r * W + j - 1 + W;       // my next batch
                if (live) {
                    load_values();
                    load_slots(kb + W);
                }
            }
        }
    }
    __syncthreads();
    if (w == 0 && live) {
#pragma unroll
        for (int g = 0; g < G; ++g)   // outputs without an optimiser keep the bias (their chains were computed and are discarded)
            out[static_cast<size_t>(row) * D + d0 + g] = ((coef.cover >> (d0 + g)) & 1ull) ? token[lane * G + g] : b0[g];
    }
}

}  // namespace

size_t predict_chain_slot_ints(int n, int trees) {
    const size_t Ts = (static_cast<size_t>(trees) / kChainU) * kChainU + kSlotPad + kChainU;
    return static_cast<size_t>(n) * Ts;
}

bool predict_chain(const PredictModel &pm, const float *obs, int F, const int32_t *cat_codes, int Fc, int n, int start_tree, int stop_tree,
                   float *out, hipStream_t s) {
    const int Tn = stop_tree - start_tree;
    if (!pm.slots || Tn <= 0 || n <= 0 || F <= 0 || !pm.coef_ok || pm.D > 64) return false;
    if (pm.oblivious ? !(pm.obl_ok && pm.cond_pack) : !(pm.grd_ok && pm.grd_nodes)) return false;
    if (static_cast<long long>(pm.n_leaves) * pm.D * 4 >= (1ll << 31)) return false;   // slots are 32-bit byte offsets into the values
    if (predict_chain_slot_ints(n, Tn) > pm.slot_ints) return false;
    if (static_cast<long long>(n) * pm.D >= (1ll << 31)) return false;                 // the relay indexes (row, output) lanes with an int
    const int xs = F | 1;
    int rows = static_cast<int>(std::min<size_t>(64, (48 * 1024) / (static_cast<size_t>(xs) * sizeof(float))));
    if (rows < 1) return false;
    // enough blocks for the chip before the row tiles grow: blocks = ceil(Ts / 256) * ceil(n / rows)
    const int Ts = static_cast<int>(predict_chain_slot_ints(1, Tn));
    const int tree_blocks = (Ts + 255) / 256;
    while (rows > 8 && static_cast<long long>(tree_blocks) * ((n + rows - 1) / rows) < 1024) rows >>= 1;
    const size_t lds = static_cast<size_t>(rows) * xs * sizeof(float);
    dim3 grid(tree_blocks, (n + rows - 1) / rows);
#define GBRL_SLOTS(G_, LV_) hipLaunchKernelGGL((k_leaf_slots<G_, LV_>), grid, dim3(256), lds, s, pm.tree_indices, pm.depths, pm.cond_pack, pm.max_depth, pm.grd_nodes, pm.grd_node_off, pm.D, obs, F, cat_codes, Fc, n, start_tree, Tn, Ts, rows, pm.slots)
    if (!pm.oblivious) GBRL_SLOTS(true, 8);
    else if (pm.max_depth <= 4) GBRL_SLOTS(false, 4);
    else if (pm.max_depth <= 6) GBRL_SLOTS(false, 6);
    else GBRL_SLOTS(false, 8);
#undef GBRL_SLOTS
    ChainCoef coef;
    for (int j = 0; j < 64; ++j) coef.c[j] = j < pm.D ? -pm.coef[j] : 0.0f;
    coef.cover = pm.coef_cover;
    const int n_lanes = n * pm.D;
    hipLaunchKernelGGL((k_chain_relay<4, 64, 1>), dim3(static_cast<unsigned>((n_lanes + 63) / 64)), dim3(256), 0, s, pm.slots, Tn, Ts, pm.values,
                       pm.bias, coef, pm.D, n_lanes, out);
    return true;
}

}  // namespace kern
}  // namespace gbrl
