// small_prep.h -- device bodies of the RL-sized preparation kernels, shared by their stand-alone launches (k_small_stats in kernels.hip,
// k_sort_quantiles in quantile.hip) and by the fused preparation kernel (k_small_prep, small_prep.hip: statistics, quantisation, split
// candidates and class codes of a step of a few thousand rows in ONE launch).  One body, so that a step computes the same bits whichever
// launch runs it.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "kernels.h"
#include "kernels_common.h"

namespace gbrl {
namespace kern {
namespace {

__device__ __forceinline__ int ilog2_floor_dev(double x) { int e; (void)frexp(x, &e); return e - 1; }

// ------------------------------------------------------------------------------------------------------------
// RL-sized batches (round 4): the whole statistics chain -- column sums, mean, centred squares, std / maxima / scales, quantisation --
// in ONE block.  The chain above takes seven launches for an L2 step; at a few thousand rows each of them is pure launch latency.
// The sums keep their BITS: the block replays the reduction tree of k_column_sums / k_column_sums_final on "virtual threads"
// (virtual block b, thread t accumulates the elements b*bs + t + k*stride in order; per (block, column) the partials are added
// in increasing t; the final kernel's 256-entry tree is run per output), so a step gives the same quantised gradients whichever
// path it takes.  Host guarantees: n_blocks * bs <= kSmallStatsVirtual, D <= 16.
// ------------------------------------------------------------------------------------------------------------
constexpr int kSmallStatsVirtual = 8192;
constexpr int kSmallStatsThreads = 1024;
__device__ __forceinline__ void small_stats_body(const float *__restrict__ g, int n, int D, int n_blocks, int bs, int centred /*L2: standardise*/,
                                                 int chunk_rows, double *__restrict__ stat /*[4D]*/, float *__restrict__ meanden /*[2D]*/,
                                                 StepScales *__restrict__ sc, int32_t *__restrict__ qg, double *sd /*dynamic LDS: [V] sums, [V] maxima, then [n_blocks][2D] partials*/) {
    const int V = n_blocks * bs;
    double *vacc = sd, *vmax = sd + V, *part = sd + 2 * V;
    __shared__ float s_center[16], s_den[16];
    __shared__ double s_stat[64];                        // the block's own copy of stat[4D] (D <= 16): no global round trip between phases
    const size_t n_el = static_cast<size_t>(n) * D;
    const size_t stride = static_cast<size_t>(V);
    auto sums = [&](bool use_center, double *out /*[2D]*/) {
        for (int v = threadIdx.x; v < V; v += kSmallStatsThreads) {
            const int col = (v % bs) % D;
            const float c = use_center ? s_center[col] : 0.0f;
            double acc = 0.0;
            float mx = 0.0f;
            // (k_column_sums: the order of the plain grid-stride loop; eight loads in flight -- a virtual thread owns at most eight
            // elements when the grid is sized like column_sums_blocks, so this is usually one batch)
            for (size_t e0 = static_cast<size_t>(v); e0 < n_el; e0 += 8 * stride) {
                float xs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const size_t e = e0 + u * stride; xs[u] = e < n_el ? g[e] : 0.0f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (e0 + u * stride < n_el) {
                        if (use_center) {
                            const float dv = xs[u] - c;
                            acc += static_cast<double>(dv) * static_cast<double>(dv);
                            mx = fmaxf(mx, fabsf(dv));
                        } else {
                            acc += static_cast<double>(xs[u]);
                            mx = fmaxf(mx, fabsf(xs[u]));
                        }
                    }
                }
            }
            vacc[v] = acc;
            vmax[v] = static_cast<double>(mx);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n_blocks * D; i += kSmallStatsThreads) {   // per (virtual block, column): increasing t
            const int b = i / D, d = i % D;
            double s2 = 0.0, m = 0.0;
            // (the additions keep their order; the operands of eight of them are loaded together -- with one output this loop is 256 dependent
            // LDS round trips on two threads otherwise: 10 of the 17 us of the statistics)
            int t = d;
            for (; t + 7 * D < bs; t += 8 * D) {
                double x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { x[u] = vacc[b * bs + t + u * D]; y[u] = vmax[b * bs + t + u * D]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s2 += x[u]; m = fmax(m, y[u]); }
            }
            for (; t < bs; t += D) { s2 += vacc[b * bs + t]; m = fmax(m, vmax[b * bs + t]); }
            part[static_cast<size_t>(b) * 2 * D + d] = s2;
            part[static_cast<size_t>(b) * 2 * D + D + d] = m;
        }
        __syncthreads();
        // k_column_sums_final, output o (D sums, then D maxima): its 256 threads fold the partials b = x, x + 256, ... (here n_blocks <= 32:
        // one term, 0.0 + partial) and run the 128 .. 1 tree.  Entries x >= n_blocks are +0.0 and no folded sum is ever -0.0, so the
        // stages 128, 64 and 32 add exact zeros: the tree is replayed from stage 16 on 32 entries, by one thread per output, in registers
        // (a block-wide replay cost ten barriers per four outputs: 33 us for 8 outputs, more than the launches it saved).
        if (threadIdx.x < 2 * D) {
            const int o = threadIdx.x;
            const bool is_max = o >= D;
            double t[32];
#pragma unroll
            for (int x = 0; x < 32; ++x) {
                const double v2 = x < n_blocks ? part[static_cast<size_t>(x) * 2 * D + o] : 0.0;
                t[x] = is_max ? fmax(0.0, v2) : 0.0 + v2;
            }
#pragma unroll
            for (int w = 16; w > 0; w >>= 1) {
#pragma unroll
                for (int x = 0; x < 16; ++x)
                    if (x < w) t[x] = is_max ? fmax(t[x], t[x + w]) : t[x] + t[x + w];
            }
            out[o] = t[0];
            s_stat[(out - stat) + o] = t[0];
        }
        __syncthreads();
    };
    sums(false, stat);
    __syncthreads();
    if (centred) {
        if (threadIdx.x < D) {   // k_stats_mean
            const float m = static_cast<float>(s_stat[threadIdx.x] / static_cast<double>(static_cast<long long>(n)));
            meanden[threadIdx.x] = m;
            s_center[threadIdx.x] = m;
        }
        __syncthreads();
        sums(true, stat + 2 * D);
        __syncthreads();
    }
    // k_stats_finish (maxima: order-free)
    __shared__ float s_h0, s_h1;
    if (threadIdx.x == 0) {
        const double *stat_raw = s_stat, *stat_centred = centred ? s_stat + 2 * D : nullptr;
        float h0 = 0.0f, h1 = 0.0f;
        const float recip = __fdiv_rn(1.0f, __fsub_rn(static_cast<float>(static_cast<long long>(n)), 1.0f));
        for (int d = 0; d < D; ++d) {
            h1 = fmaxf(h1, static_cast<float>(stat_raw[D + d]));
            if (!(fabs(stat_raw[d]) < INFINITY)) h1 = INFINITY;
            if (stat_centred && !(fabs(stat_centred[d]) < INFINITY)) h0 = INFINITY;
            if (stat_centred) {
                const float sdv = __fsqrt_rn(__fmul_rn(static_cast<float>(stat_centred[d]), recip));
                const float den = __fadd_rn(sdv, 1e-8f);
                meanden[D + d] = den;
                s_den[d] = den;
                h0 = fmaxf(h0, __fmul_rn(__fdiv_rn(static_cast<float>(stat_centred[D + d]), den), 1.0001f));
            }
        }
        if (h0 != h0) h0 = INFINITY;
        if (h1 != h1) h1 = INFINITY;
        const float hraw = h1, hbuild = stat_centred ? h0 : h1;
        int sbits = 20, lbits = 40;
        if (hbuild > 0.f && hbuild < INFINITY) sbits = min(100, ilog2_floor_dev(2147483647.0 / (static_cast<double>(chunk_rows) * hbuild)) - 1);
        if (hraw > 0.f && hraw < INFINITY) lbits = min(60, ilog2_floor_dev(4.0e18 / (static_cast<double>(static_cast<long long>(n)) * hraw)) - 1);
        StepScales o{};
        o.sbits = sbits; o.lbits = lbits;
        o.scale = static_cast<float>(ldexp(1.0, sbits));
        o.inv_scale = ldexp(1.0, -sbits);
        o.leaf_scale = ldexp(1.0, lbits);
        o.hmax_build = hbuild; o.hmax_raw = hraw;
        *sc = o;
        s_h0 = o.scale;
    }
    __syncthreads();
    __threadfence_block();
    const float scale = s_h0;
    // k_quantize
    for (size_t e = threadIdx.x; e < n_el; e += kSmallStatsThreads) {
        const int col = static_cast<int>(e % D);
        float v = g[e];
        if (centred) v = (v - s_center[col]) / s_den[col];
        qg[e] = __float2int_rn(v * scale);
    }
    (void)s_h1;
}


// x of lane (lane ^ M), M = 1 .. 32, as VALU data movement: quad permutations (1, 2), mirrors composed (4 = half-mirror then quad reversal:
// ^7 ^3; 8 = row mirror then half-mirror: ^15 ^7), v_permlane16_swap / v_permlane32_swap (gfx950) for 16 / 32.
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t x) {
    const int v = static_cast<int>(x);
    if constexpr (M == 1) return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));          // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true), 0x1B, 0xf, 0xf, true));
    else if constexpr (M == 8) return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true), 0x141, 0xf, 0xf, true));
    else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // r[0]: odd rows hold the even rows' values; r[1]: even rows hold the odd rows'
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // r[0]: upper half holds the lower half's values; r[1]: the reverse
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
// out = mask bit of the lane ? if_set : if_clear -- one v_cndmask with the 64-bit lane mask in scalar registers
__device__ __forceinline__ uint32_t select_by_lane_mask(uint32_t if_clear, uint32_t if_set, unsigned long long mask) {
    uint32_t out;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(if_clear), "v"(if_set), "s"(mask));
    return out;
}
// Lanes whose elements i = 256 * wave + 4 * lane + r have (i & V) == 0, V a power of two >= 4: a constant below 256 (a lane bit), the
// whole wave or none of it from 256 on (a wave bit).
template <int V>
__device__ __forceinline__ unsigned long long bitonic_zero_mask(int wave_u) {
    static_assert(V >= 4 && (V & (V - 1)) == 0, "a power of two >= 4");
    if constexpr (V >= 256) return ((wave_u * 256) & V) == 0 ? ~0ull : 0ull;
    else if constexpr (V == 4) return ~0xAAAAAAAAAAAAAAAAull;
    else if constexpr (V == 8) return ~0xCCCCCCCCCCCCCCCCull;
    else if constexpr (V == 16) return ~0xF0F0F0F0F0F0F0F0ull;
    else if constexpr (V == 32) return ~0xFF00FF00FF00FF00ull;
    else if constexpr (V == 64) return ~0xFFFF0000FFFF0000ull;
    else return ~0xFFFFFFFF00000000ull;
}
// One compare-exchange stage of the bitonic network: run length K, distance J; four keys per thread (element i = 256 * wave + 4 * lane + r).
//   J >= 256   partners sit in other waves: through LDS (two barriers); the whole wave takes minima or maxima
//   4 .. 128   partners sit in other lanes: DPP / permlane moves (lane_xor), the lanes that keep the minimum are a constant 64-bit mask
//              ((i & J) == 0) == ((i & K) == 0) -- one v_cndmask per key, no integer arithmetic on i
//   1, 2       partners sit in the thread's own registers (static pairs)
template <int K, int J>
__device__ __forceinline__ void bitonic_stage(uint32_t (&a)[4], uint32_t *s, int base, bool act, int wave_u) {
    if constexpr (J >= 256) {
        if (act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[base + r] = a[r];
        }
        __syncthreads();
        const bool take_min = (((wave_u * 256) & J) == 0) == (((wave_u * 256) & K) == 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t other = act ? s[(base + r) ^ J] : 0u;
            a[r] = take_min ? min(a[r], other) : max(a[r], other);
        }
        __syncthreads();
    } else if constexpr (J >= 4) {
        const unsigned long long tm = ~(bitonic_zero_mask<J>(wave_u) ^ bitonic_zero_mask<K>(wave_u));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t other = lane_xor<(J >> 2)>(a[r]);
            a[r] = select_by_lane_mask(max(a[r], other), min(a[r], other), tm);
        }
    } else if constexpr (K >= 4) {
        const unsigned long long up = bitonic_zero_mask<K>(wave_u);       // ascending where (i & K) == 0
        auto cx = [&](uint32_t &x, uint32_t &y) {
            const uint32_t lo = min(x, y), hi = max(x, y);
            x = select_by_lane_mask(hi, lo, up);
            y = select_by_lane_mask(lo, hi, up);
        };
        if constexpr (J == 2) { cx(a[0], a[2]); cx(a[1], a[3]); }
        else { cx(a[0], a[1]); cx(a[2], a[3]); }
    } else {   // K == 2, J == 1: (i & 2) == 0 for the pair (0, 1), != 0 for (2, 3)
        const uint32_t lo0 = min(a[0], a[1]), hi0 = max(a[0], a[1]), lo1 = min(a[2], a[3]), hi1 = max(a[2], a[3]);
        a[0] = lo0; a[1] = hi0; a[2] = hi1; a[3] = lo1;
    }
}
// the stages from (K, J) to the end of the network for S keys, in order: J = K/2 .. 1 for K = 2, 4, .. S
template <int S, int K, int J>
__device__ __forceinline__ void bitonic_from(uint32_t (&a)[4], uint32_t *s, int base, bool act, int wave_u) {
    bitonic_stage<K, J>(a, s, base, act, wave_u);
    if constexpr (J > 1) bitonic_from<S, K, (J >> 1)>(a, s, base, act, wave_u);
    else if constexpr (K < S) bitonic_from<S, K * 2, K>(a, s, base, act, wave_u);
}

// Class codes of one feature against thresholds staged in LDS (sorted ascending): code = #{k : thr_s[k] < key}, by the same descent as
// sort_quantiles_body's.
__device__ __forceinline__ int code_of_key(const uint32_t *thr_s, int B, int top, uint32_t key) {
    int pos = 0;
    for (int step = top >> 1; step > 0; step >>= 1) {
        const int np = pos + step;
        if (np <= B && thr_s[np - 1] < key) pos = np;
    }
    return pos;
}

// ---- small batches (RL-sized): the whole column fits in LDS -> sort it, read the ranks ---------------------------------------
// One block per feature: n <= S keys padded with the maximal key to S (a power of two <= 16384), bitonic sort in LDS,
// thr_keys[f][k] = sorted[cum[k] - 1].  One launch instead of the eight of the radix multi-select, which are launch-bound at
// these sizes.  Exact by construction (padding sorts last and no rank points into it).
__device__ __forceinline__ float key_to_float_q(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
// `codes` (nullable; round 4): the block also writes its feature's class codes -- codes[g][row][fl] = #{k : thr_key[f][k] < key(row, f)},
// what k_bin_cols computes -- from the thresholds it has just selected (a binary search per row in LDS), so an RL-sized step needs no
// separate binning launch.  The grid then covers the padding features of the last group of 16 too: their blocks write zeros.
// ROWMAJOR: `src` is the caller's row-major float matrix obs[n][F] (the fused preparation kernel: no transposed key matrix exists); else the
// feature-major key matrix kt[F][n].  Threads beyond S / 4 (the fused kernel runs 1024 per block whatever S is) only keep the barriers company.
template <bool ROWMAJOR>
__device__ __forceinline__ void sort_quantiles_body(const void *__restrict__ src, int n, int S, const int64_t *__restrict__ cum, int B,
                                                    uint32_t *__restrict__ thr_keys, float *__restrict__ thr_floats, int F, int f,
                                                    uint16_t *__restrict__ codes, uint32_t *s /*dynamic LDS: [S] keys, then [B] selected thresholds*/,
                                                    uint32_t *prof = nullptr /*measurement: feature 0's time per stage, 10 ns units*/,
                                                    uint16_t *__restrict__ codes_fm = nullptr /*nullable: a second, feature-major copy [F][n] of the codes
                                                    (the one-launch growth kernel reads a slot's column and routes rows with coalesced loads)*/) {
    long long pt = prof ? wall_clock64() : 0;
    auto mark = [&](int i) { if (prof && f == 0 && threadIdx.x == 0) { const long long n_ = wall_clock64(); prof[i] = static_cast<uint32_t>(n_ - pt); pt = n_; } };
    if (f >= F) {                     // padding feature of the last code group (codes != nullptr only)
        uint16_t *dst = codes + (static_cast<size_t>(f >> 4) * n) * kCodeGroup + (f & (kCodeGroup - 1));
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[static_cast<size_t>(i) * kCodeGroup] = 0;
        return;
    }
    const uint32_t *col = ROWMAJOR ? nullptr : static_cast<const uint32_t *>(src) + static_cast<size_t>(f) * n;
    const float *ocol = ROWMAJOR ? static_cast<const float *>(src) + f : nullptr;
    auto key_at = [&](int i) -> uint32_t { return ROWMAJOR ? float_to_key(ocol[static_cast<size_t>(i) * F]) : col[i]; };
    // Bitonic sort, four keys per thread (blockDim.x = S / 4, S >= 256): element i = 256 * wave + 4 * lane + r.  Compare-exchange
    // distances 1 and 2 stay inside a thread's registers, 4 .. 128 are lane exchanges inside the wave (no LDS memory, no barrier), and
    // only the distances >= 256 cross waves through LDS (10 of the 78 stages at S = 4096).  Round 3 kept every key in LDS and paid two
    // dependent LDS round trips per stage: 31 us for 4096 keys, the largest kernel of an RL-sized step.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = wave * 256 + lane * 4;
    const bool act = base < S;        // (whole waves: 256 keys per wave)
    uint32_t a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = base + r < n ? key_at(base + r) : 0xffffffffu;
    const uint32_t orig[4] = {a[0], a[1], a[2], a[3]};
    if (prof) { __syncthreads(); mark(0); }
    // The 78 compare-exchange stages of a 4096-key sort, unrolled at compile time per S (bitonic_from): distances, directions and lane
    // masks are constants, no loop counters and no branches between the stages.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    switch (S) {
        case 256: bitonic_from<256, 2, 1>(a, s, base, act, wave_u); break;
        case 512: bitonic_from<512, 2, 1>(a, s, base, act, wave_u); break;
        case 1024: bitonic_from<1024, 2, 1>(a, s, base, act, wave_u); break;
        case 2048: bitonic_from<2048, 2, 1>(a, s, base, act, wave_u); break;
        default: bitonic_from<4096, 2, 1>(a, s, base, act, wave_u); break;
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[base + r] = a[r];
    }
    __syncthreads();
    mark(1);
    // keys and floats at once (what k_keys_to_floats does for the other selections: a key in the NaN range is raised to -inf's key)
    uint32_t *thr_s = s + S;
    for (int k = threadIdx.x; k < B; k += blockDim.x) {
        uint32_t key = s[cum[k] - 1];
        if (key < 0x007fffffu) key = 0x007fffffu;
        thr_keys[static_cast<size_t>(f) * B + k] = key;
        thr_floats[static_cast<size_t>(f) * B + k] = key_to_float_q(key);
        if (codes) thr_s[k] = key;
    }
    if (codes == nullptr) return;
    __syncthreads();
    mark(2);
    int top = 1;
    while (top <= B) top <<= 1;                      // 2^m > B: the descent can reach every count 0 .. B
    uint16_t *dst = codes + (static_cast<size_t>(f >> 4) * n) * kCodeGroup + (f & (kCodeGroup - 1));
    // (the thread's own four keys, kept from the load: no second pass over the column)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = base + r;
        if (i < n) {
            const int pos = code_of_key(thr_s, B, top, orig[r]);
            dst[static_cast<size_t>(i) * kCodeGroup] = static_cast<uint16_t>(pos);
            if (codes_fm) codes_fm[static_cast<size_t>(f) * n + i] = static_cast<uint16_t>(pos);
        }
    }
    if (prof) { __syncthreads(); mark(3); }
}


// A4 in one block per feature (uniform candidates, split_candidate_generator.cpp:59-76): column minimum / maximum of the finite keys (what
// k_column_minmax finds), the thresholds min + b * step as ONE fma (k_uniform_thresholds, Q5), and the feature's class codes.
// LDS: [n] keys, then [B] threshold keys, then 2 x 16 words of reduction scratch.
__device__ __forceinline__ void uniform_thresholds_body(const float *__restrict__ obs, int n, int F, int f, int B, float *__restrict__ thr,
                                                        uint32_t *__restrict__ thr_keys, uint16_t *__restrict__ codes, uint32_t *lds,
                                                        uint16_t *__restrict__ codes_fm = nullptr) {
    uint16_t *dst = codes + (static_cast<size_t>(f >> 4) * n) * kCodeGroup + (f & (kCodeGroup - 1));
    if (f >= F) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[static_cast<size_t>(i) * kCodeGroup] = 0;
        return;
    }
    uint32_t *keys = lds, *thr_s = lds + n, *red = lds + n + B;
    constexpr uint32_t kMinFinite = 0x007fffffu;   // key of -inf; smaller keys are NaNs, which the reference's `<` / `>` scan never picks up
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t k = float_to_key(obs[static_cast<size_t>(i) * F + f]);
        keys[i] = k;
        if (k >= kMinFinite) { lo = min(lo, k); hi = max(hi, k); }
    }
    for (int o = kWave / 2; o > 0; o >>= 1) {
        lo = min(lo, static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo), o, kWave)));
        hi = max(hi, static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi), o, kWave)));
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, n_waves = (blockDim.x + kWave - 1) / kWave;
    if (lane == 0) { red[wave] = lo; red[16 + wave] = hi; }
    __syncthreads();
    lo = 0xffffffffu; hi = 0u;
    for (int w = 0; w < n_waves; ++w) { lo = min(lo, red[w]); hi = max(hi, red[16 + w]); }
    const float flo = key_to_float(lo), fhi = key_to_float(hi);
    const float step = (fhi - flo) / static_cast<float>(B);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float t = fmaf(static_cast<float>(b), step, flo);
        if (t != t) t = -INFINITY;       // (k_uniform_thresholds: a column with infinite values)
        const uint32_t tk = float_to_key(t);
        thr[static_cast<size_t>(f) * B + b] = t;
        thr_keys[static_cast<size_t>(f) * B + b] = tk;
        thr_s[b] = tk;
    }
    __syncthreads();
    int top = 1;
    while (top <= B) top <<= 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint16_t cv = static_cast<uint16_t>(code_of_key(thr_s, B, top, keys[i]));
        dst[static_cast<size_t>(i) * kCodeGroup] = cv;
        if (codes_fm) codes_fm[static_cast<size_t>(f) * n + i] = cv;
    }
}

}  // namespace
}  // namespace kern
}  // namespace gbrl
