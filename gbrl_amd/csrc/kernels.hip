// kernels.hip -- gfx950 (CDNA4, wave64) kernels behind GBRL::step: gradient statistics, uniform thresholds, histogram build /
// reduce, split scores and selection, row partition, leaf sums.  (Prediction: predict.hip; categorical cells: categorical.hip;
// quantile thresholds: radix_select.hip, quantile.hip.)
//
// Arithmetic policy (DESIGN.md "exact sums"): every sum that feeds a split decision or a leaf value is an INTEGER
// sum of fixed-point values, so results do not depend on the order in which waves, blocks or GPUs add them
// (deterministic, and bit-identical for 1/2/4/8 row shards).  Scores are evaluated in fp64 from those exact sums and
// rounded to fp32 once, then compared the way the reference compares them (fitter.cpp:332-341, 435-444).
#include "kernels.h"
#include "hooks.h"
#include "kernels_common.h"
#include "score_common.h"
#include "small_prep.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <cstdlib>

namespace gbrl {
namespace kern {

namespace {

// ------------------------------------------------------------------------------------------------------------
// A2  gradient statistics: out[d] = sum_i (g[i,d] - center[d])^2  (center != null)  or  sum_i g[i,d]
// Block size is a multiple of D so that every thread owns one column; per-block partials are combined by a single
// block in a fixed order => deterministic.
// ------------------------------------------------------------------------------------------------------------
__global__ void k_column_sums(const float *__restrict__ g, size_t n_el, int D, const float *__restrict__ center,
                              double *__restrict__ partials /*[blocks][2D]: sums then maxima*/) {
    extern __shared__ double sh[];  // [2*bs]
    const int bs = blockDim.x;
    const int col = threadIdx.x % D;
    const float c = center ? center[col] : 0.0f;
    double acc = 0.0;
    float mx = 0.0f;
    auto take = [&](float v) {
        if (center) {
            const float dv = v - c;  // fp32 subtraction like the reference (math_ops.cpp:498)
            acc += static_cast<double>(dv) * static_cast<double>(dv);
            mx = fmaxf(mx, fabsf(dv));
        } else {
            acc += static_cast<double>(v);
            mx = fmaxf(mx, fabsf(v));
        }
    };
    // eight loads in flight per thread, accumulated in the order of the plain grid-stride loop (the sums keep their bits)
    const size_t stride = static_cast<size_t>(gridDim.x) * bs;
    size_t e = static_cast<size_t>(blockIdx.x) * bs + threadIdx.x;
    for (; e + 7 * stride < n_el; e += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[e + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) take(v[u]);
    }
    for (; e < n_el; e += stride) take(g[e]);
    sh[threadIdx.x] = acc;
    sh[bs + threadIdx.x] = static_cast<double>(mx);
    __syncthreads();
    if (threadIdx.x < D) {
        double s = 0.0, m = 0.0;
        for (int t = threadIdx.x; t < bs; t += D) { s += sh[t]; m = fmax(m, sh[bs + t]); }
        partials[static_cast<size_t>(blockIdx.x) * 2 * D + threadIdx.x] = s;
        partials[static_cast<size_t>(blockIdx.x) * 2 * D + D + threadIdx.x] = m;
    }
}
__global__ __launch_bounds__(256) void k_column_sums_final(const double *__restrict__ partials, int n_blocks, int D,
                                                           double *__restrict__ out /*[2D]*/) {
    // one block per output (D sums, then D maxima); fixed-shape tree => deterministic
    __shared__ double sh[256];
    const int d = blockIdx.x;
    const bool is_max = d >= D;
    double s = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) {
        const double v = partials[static_cast<size_t>(b) * 2 * D + d];
        s = is_max ? fmax(s, v) : s + v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] = is_max ? fmax(sh[threadIdx.x], sh[threadIdx.x + o]) : sh[threadIdx.x] + sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[d] = sh[0];
}

// Statistics chain on the device, operation for operation what the host path does (fitter.cpp:58-63, math_ops.cpp:464,510,94)
// with explicitly rounded fp32 intrinsics (no contraction).
__global__ __launch_bounds__(256) void k_stats_mean(const double *__restrict__ stat, long long n, int D, float *__restrict__ meanden) {
    for (int d = threadIdx.x; d < D; d += blockDim.x) meanden[d] = static_cast<float>(stat[d] / static_cast<double>(n));
}
__global__ __launch_bounds__(256) void k_stats_finish(const double *__restrict__ stat_raw, const double *__restrict__ stat_centred,
                                                      long long n, int D, int chunk_rows, float *__restrict__ meanden,
                                                      StepScales *__restrict__ sc) {
    __shared__ float m0[256], m1[256];
    float h0 = 0.0f, h1 = 0.0f;
    const float recip = __fdiv_rn(1.0f, __fsub_rn(static_cast<float>(n), 1.0f));
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        h1 = fmaxf(h1, static_cast<float>(stat_raw[D + d]));
        if (!(fabs(stat_raw[d]) < INFINITY)) h1 = INFINITY;            // a NaN / inf gradient poisons its column sum
        if (stat_centred && !(fabs(stat_centred[d]) < INFINITY)) h0 = INFINITY;
        if (stat_centred) {
            const float sd = __fsqrt_rn(__fmul_rn(static_cast<float>(stat_centred[d]), recip));
            const float den = __fadd_rn(sd, 1e-8f);
            meanden[D + d] = den;
            h0 = fmaxf(h0, __fmul_rn(__fdiv_rn(static_cast<float>(stat_centred[D + d]), den), 1.0001f));
        }
    }
    // NaN must survive the reduction (fmaxf drops it): carried as +inf, which the host rejects
    if (h0 != h0) h0 = INFINITY;
    if (h1 != h1) h1 = INFINITY;
    m0[threadIdx.x] = h0;
    m1[threadIdx.x] = h1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { m0[threadIdx.x] = fmaxf(m0[threadIdx.x], m0[threadIdx.x + o]); m1[threadIdx.x] = fmaxf(m1[threadIdx.x], m1[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float hraw = m1[0], hbuild = stat_centred ? m0[0] : m1[0];
        int sbits = 20, lbits = 40;
        if (hbuild > 0.f && hbuild < INFINITY) sbits = min(100, ilog2_floor_dev(2147483647.0 / (static_cast<double>(chunk_rows) * hbuild)) - 1);
        if (hraw > 0.f && hraw < INFINITY) lbits = min(60, ilog2_floor_dev(4.0e18 / (static_cast<double>(n) * hraw)) - 1);
        StepScales o{};
        o.sbits = sbits; o.lbits = lbits;
        o.scale = static_cast<float>(ldexp(1.0, sbits));
        o.inv_scale = ldexp(1.0, -sbits);
        o.leaf_scale = ldexp(1.0, lbits);
        o.hmax_build = hbuild; o.hmax_raw = hraw;
        *sc = o;
    }
}

// RL-sized batches (round 4): the whole statistics chain in ONE block -- small_prep.h, small_stats_body (shared with the fused
// preparation kernel of small_prep.hip).
__global__ __launch_bounds__(kSmallStatsThreads) void k_small_stats(const float *__restrict__ g, int n, int D, int n_blocks, int bs, int centred /*L2: standardise*/,
                                                                    int chunk_rows, double *__restrict__ stat /*[4D]*/, float *__restrict__ meanden /*[2D]*/,
                                                                    StepScales *__restrict__ sc, int32_t *__restrict__ qg) {
    extern __shared__ double sd[];                       // [V] sums, [V] maxima, then [n_blocks][2D] partials
    small_stats_body(g, n, D, n_blocks, bs, centred, chunk_rows, stat, meanden, sc, qg, sd);
}

__device__ __forceinline__ float standardise(float v, const float *mean, const float *denom, int col) {
    // (g - mean) / (std + 1e-8f) evaluated in fp32 exactly like fitter.cpp:58-63 -> math_ops.cpp:498,94
    if (mean == nullptr) return v;
    return (v - mean[col]) / denom[col];
}

__global__ void k_quantize(const float *__restrict__ g, size_t n_el, int D, const float *__restrict__ mean,
                           const float *__restrict__ denom, const StepScales *__restrict__ sc, int32_t *__restrict__ qg) {
    const float scale = sc->scale;
    for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n_el;
         e += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const float v = standardise(g[e], mean, denom, static_cast<int>(e % D));
        qg[e] = __float2int_rn(v * scale);  // scale is a power of two: the product is exact, one rounding to integer
    }
}

// ------------------------------------------------------------------------------------------------------------
// A4  uniform candidates: per-column min / max (order independent => atomics are deterministic)
// ------------------------------------------------------------------------------------------------------------
// Column minima / maxima from the feature-major keys (the transpose has already run: every later pass streams columns).  One block
// per (row chunk, feature): 16-byte loads, eight in flight per thread, wave reduction, one atomic pair per wave.
__global__ __launch_bounds__(256) void k_column_minmax(const uint32_t *__restrict__ kt, int n, int rows_per_block, uint32_t *__restrict__ mn,
                                                       uint32_t *__restrict__ mx) {
    const int f = blockIdx.y;
    const uint32_t *col = kt + static_cast<size_t>(f) * n;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    uint32_t lo = 0xffffffffu, hi = 0u;
    constexpr uint32_t kMinFinite = 0x007fffffu;   // key of -inf; smaller keys are NaNs, which the reference's `<` / `>` scan never picks up
    auto upd = [&](uint32_t k) { if (k >= kMinFinite) { lo = min(lo, k); hi = max(hi, k); } };
    // the column start is 4-byte aligned only (n is arbitrary): peel to a 16-byte boundary, then vector loads
    int r = r0 + threadIdx.x;
    const uintptr_t mis = (reinterpret_cast<uintptr_t>(col + r0) >> 2) & 3;
    const int head = min(r1 - r0, static_cast<int>((4 - mis) & 3));
    if (threadIdx.x < head) upd(col[r]);
    const int v0 = r0 + head, nvec = (r1 - v0) / 4;
    const uint4 *vec = reinterpret_cast<const uint4 *>(col + v0);
    for (int i = threadIdx.x; i < nvec; i += 256 * 4) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = vec[min(i + u * 256, nvec - 1)];   // past the end: re-read the last vector (harmless for min / max)
#pragma unroll
        for (int u = 0; u < 4; ++u) { upd(q[u].x); upd(q[u].y); upd(q[u].z); upd(q[u].w); }
    }
    for (r = v0 + nvec * 4 + threadIdx.x; r < r1; r += 256) upd(col[r]);
    for (int o = kWave / 2; o > 0; o >>= 1) {
        lo = min(lo, static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo), o, kWave)));
        hi = max(hi, static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi), o, kWave)));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMin(&mn[f], lo);
        atomicMax(&mx[f], hi);
    }
}
__global__ void k_uniform_thresholds(const uint32_t *__restrict__ mn, const uint32_t *__restrict__ mx, int F, int B,
                                     float *__restrict__ thr, uint32_t *__restrict__ thr_keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * B) return;
    const int f = i / B, b = i % B;
    const float lo = key_to_float(mn[f]), hi = key_to_float(mx[f]);
    const float step = (hi - lo) / static_cast<float>(B);
    // split_candidate_generator.cpp:67: min + b*step, contracted to ONE fma by the reference's release build (Q5)
    float t = fmaf(static_cast<float>(b), step, lo);
    // a column with infinite values gives inf - inf = NaN thresholds (in the reference too): a NaN threshold never passes `x > t`;
    // -inf is stored instead so that the key comparison of step() (NaN has no key above it) and predict()'s float comparison agree
    if (t != t) t = -INFINITY;
    thr[i] = t;
    if (thr_keys) thr_keys[i] = float_to_key(t);   // (what a k_floats_to_keys launch behind this one did)
}

// ------------------------------------------------------------------------------------------------------------
// A3  binning / counting:  j(row, f) = #{k : trial[f][k] (<|<=) key(obs[row,f])},  counts[f][j] += 1,
//     optionally codes[row][f] = j.  Used (a) by the exact quantile bisection (non-strict, counts only) and
//     (b) to turn observations into bin codes once thresholds are known (strict: code = #{k : t_k < x}).
// One block = FT features x (256/FT) rows per iteration; trial keys staged in LDS transposed ([k][f], f fastest)
// so that the lanes of a wave, which hold consecutive features, always hit distinct banks.
// ------------------------------------------------------------------------------------------------------------
constexpr int kBinThreads = 256;

// FT = features per block tile: 64 (256 B of each row) while the 2 B + 1 LDS rows of a feature fit, fewer features for more thresholds
template <bool STRICT, int FT>
__global__ __launch_bounds__(kBinThreads) void k_bin_rows(const float *__restrict__ obs, int n, int F,
                                                           const uint32_t *__restrict__ trial, int B,
                                                           unsigned long long *__restrict__ counts, uint16_t *__restrict__ codes,
                                                           int code_stride, int code_off) {
    extern __shared__ uint32_t lds[];
    uint32_t *t = lds;                    // [B][FT]
    uint32_t *cnt = lds + B * FT;     // [B+1][FT]
    const int f0 = blockIdx.y * FT;
    const int nf = min(FT, F - f0);
    for (int i = threadIdx.x; i < B * FT; i += kBinThreads) {
        const int k = i / FT, fl = i % FT;
        t[i] = fl < nf ? trial[static_cast<size_t>(f0 + fl) * B + k] : 0xffffffffu;
    }
    if (counts)
        for (int i = threadIdx.x; i < (B + 1) * FT; i += kBinThreads) cnt[i] = 0;
    __syncthreads();
    const int fl = threadIdx.x % FT;
    const int rsub = threadIdx.x / FT;
    constexpr int RPI = kBinThreads / FT;
    if (fl < nf) {
        for (int r = blockIdx.x * RPI + rsub; r < n; r += gridDim.x * RPI) {
            const uint32_t key = float_to_key(obs[static_cast<size_t>(r) * F + f0 + fl]);
            int lo = 0, hi = B;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint32_t tv = t[mid * FT + fl];
                const bool below = STRICT ? (tv < key) : (tv <= key);
                if (below) lo = mid + 1; else hi = mid;
            }
            if (counts) atomicAdd(&cnt[lo * FT + fl], 1u);
            if (codes) codes[static_cast<size_t>(r) * code_stride + code_off + f0 + fl] = static_cast<uint16_t>(lo);
        }
    }
    if (counts) {
        __syncthreads();
        for (int i = threadIdx.x; i < (B + 1) * FT; i += kBinThreads) {
            const int j = i / FT, f = i % FT;
            const uint32_t c = cnt[i];
            if (c != 0 && f < nf) atomicAdd(&counts[static_cast<size_t>(f0 + f) * (B + 1) + j], static_cast<unsigned long long>(c));
        }
    }
}

// Exact multi-rank selection by bisection on the 32-bit ordered key (A3).  For every (feature, target k) we build the
// answer v_k = smallest key with #{keys <= v_k} >= cum_k from the most significant bit down: with prefix p_k (decided
// high bits, low bits zero) and trial t_k = p_k | bit, the bit is set iff #{keys < t_k} < cum_k.  Targets are sorted
// by rank, so prefixes and trials stay sorted per feature and one binary search per element serves all targets.
__global__ void k_qsel_init(uint32_t *__restrict__ prefix, uint32_t *__restrict__ trial, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    prefix[i] = 0u;
    trial[i] = 0x80000000u;
}
__global__ void k_qsel_update(uint32_t *__restrict__ prefix, uint32_t *__restrict__ trial,
                              const int64_t *__restrict__ counts, const int64_t *__restrict__ cum, int B, int bit,
                              int next_bit) {
    // one block per feature; thread k owns target k (loop if B > blockDim)
    extern __shared__ unsigned long long below[];  // [B+1] inclusive prefix of counts
    const int f = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int j = 0; j <= B; ++j) {
            run += static_cast<unsigned long long>(counts[static_cast<size_t>(f) * (B + 1) + j]);
            below[j] = run;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < B; k += blockDim.x) {
        const size_t i = static_cast<size_t>(f) * B + k;
        uint32_t p = prefix[i];
        // #{keys < t_k} = sum_{j<=k'} counts[j] where k' = index of the LAST trial equal to t_k ... trials are sorted and
        // counts[j] holds rows with exactly j trials <= key, so rows with key < t_k are those with j <= (#trials < t_k).
        // With duplicated trials (equal targets) the rows sit in the bucket of the first duplicate; use that index.
        int first = k;
        const uint32_t tk = trial[i];
        while (first > 0 && trial[i - (k - first) - 1] == tk) --first;
        const long long c_below = static_cast<long long>(below[first]);
        if (c_below < cum[k]) p |= (1u << bit);
        prefix[i] = p;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < B; k += blockDim.x) {
        const size_t i = static_cast<size_t>(f) * B + k;
        trial[i] = next_bit >= 0 ? (prefix[i] | (1u << next_bit)) : prefix[i];
    }
}

// Threshold keys -> floats.  A selected key in the NaN range (a column with more NaNs than one quantile step) is raised to -inf's key,
// in place, so that the key comparison of step() and the float comparison of predict() keep agreeing: `x > -inf`.
__global__ void k_keys_to_floats(uint32_t *__restrict__ keys, float *__restrict__ out, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k = keys[i];
    if (k < 0x007fffffu) { k = 0x007fffffu; keys[i] = k; }
    out[i] = key_to_float(k);
}
__global__ void k_floats_to_keys(const float *__restrict__ in, uint32_t *__restrict__ keys, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = float_to_key(in[i]);
}
__global__ void k_iota(int32_t *__restrict__ rows, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = i;
}

// ------------------------------------------------------------------------------------------------------------
// A6  split-score histogram build (THE hot kernel).
// For one chunk of one node's row list and one group of FG features, accumulate per (class, feature):
//   count and the D fixed-point gradient sums, in LDS, with ds_add_u32 atomics.
// LDS layout h[(class*(D+1) + d)*FG + f_local]: the feature index is the fastest dimension, and the 16 lanes that
// share a row hold 16 different features, so the bank of an access is decided by the lane, not by the (random)
// class: a wave-wide atomic touches each bank at most twice (two rows per 32-lane group) -- the 2-way case the LDS
// absorbs for free.  Integer adds wrap mod 2^32; the host picks the fixed-point scale so that every true partial
// sum fits in int32, which makes the wrapped result exact.
// ------------------------------------------------------------------------------------------------------------
constexpr int kHistThreads = 1024;


// Broadcast lane `SRC` of every 16-lane row to the whole row (v_mov_b32_dpp row_newbcast: a VALU move, no LDS traffic).
template <int SRC>
__device__ __forceinline__ int row_bcast(int v) {
    // every lane of the row receives lane SRC's value, so there is no "old" value to keep: mov_dpp leaves the destination
    // untied (update_dpp(0, ...) made the compiler zero 64 registers per loop iteration)
    return __builtin_amdgcn_mov_dpp(v, 0x150 + SRC, 0xf, 0xf, true);
}
template <int DT, int K>
struct RowAtomics {   // adds q[K..DT) of this lane's row into dst, one LDS atomic per output dim
    static __device__ __forceinline__ void run(int32_t *dst, int FG, int myq) {
        atomicAdd(dst + K * FG, row_bcast<K>(myq));
        RowAtomics<DT, K + 1>::run(dst, FG, myq);
    }
};
template <int DT>
struct RowAtomics<DT, DT> {
    static __device__ __forceinline__ void run(int32_t *, int, int) {}
};

// k_hist_build<DT, U>.  DT = compile-time output_dim (1..16) for the FG == 16 layout, or 0 = generic (run-time D, any FG).
// Fast path: the 16 lanes that share a row are one DPP row.  Lane fl < DT loads ONE dword, q[fl], of the row's quantised
// gradients (32 contiguous bytes per row for D = 8) and the D values are handed to all 16 lanes with row_newbcast moves,
// instead of every lane loading all D values (which made the vector-memory path, not the LDS atomics, the bottleneck).
// Every row-slot keeps U rows in flight: all loads of the U rows are issued before the first atomic.
// `Ld`: how a lane fetches its row's class code and gradient word (the product's loads below; the stand-alone harness
// scripts/hist_bench.hip instantiates the kernel with experiment loaders of its own).
struct HistLoads {
    static __device__ __forceinline__ int code(const char *cgroup, uint32_t row, uint32_t coff) {
        return *reinterpret_cast<const uint16_t *>(cgroup + (row * (kCodeGroup * 2u) + coff));
    }
    template <int DT>
    static __device__ __forceinline__ int grad(const char *qbase, uint32_t row, uint32_t qoff) {
        return *reinterpret_cast<const int32_t *>(qbase + (row * static_cast<uint32_t>(DT * 4) + qoff));
    }
};
// COUNT = false (root level, one GPU, radix-selected quantile candidates): the row count of a class is not accumulated -- eight LDS atomics
// per (row, feature) instead of nine at D = 8; kern::hist_reduce fills the count field from the selection's ranks (root_le).
template <int DT, int U, bool PIPE, class Ld = HistLoads, bool COUNT = true>
__global__ __launch_bounds__(kHistThreads) void k_hist_build(const uint16_t *__restrict__ codes, int n_rows,
                                                              const int32_t *__restrict__ qg, int D_rt,
                                                              const int32_t *__restrict__ rows,
                                                              const Chunk *__restrict__ chunks, int n_chunks, int n_groups,
                                                              int FG, int fg_shift, int NB, int32_t *__restrict__ partials,
                                                              HistDirect direct = HistDirect{}) {
    extern __shared__ int32_t h[];
    const int D = DT ? DT : D_rt;
    if (DT) { FG = 16; fg_shift = 4; }
    // XCD-aware block -> (chunk, group) map: the blocks that share a chunk (one per feature group) re-read the same rows'
    // gradients and row ids.  Block b runs on XCD b % 8 (observed; used for speed only), so all groups of a chunk are
    // given ids that are congruent mod 8 and adjacent in launch order: the re-reads are served by that XCD's L2.
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int g = jj % n_groups;
    const int chunk_id = (jj / n_groups) * 8 + xcd;
    if (chunk_id >= n_chunks) return;
    const Chunk ck = chunks[chunk_id];
    if (ck.len <= 0 && !direct.hist) return;   // block-uniform: an unused entry of a device-planned chunk table (direct: an empty node's zeros)
    const int n_acc = NB * (D + 1) * FG;
    if ((n_acc & 3) == 0) {   // 16-byte LDS writes (FG = 16: always)
        int4 *h4 = reinterpret_cast<int4 *>(h);
        for (int i = threadIdx.x; i < n_acc / 4; i += kHistThreads) h4[i] = make_int4(0, 0, 0, 0);
    } else {
        for (int i = threadIdx.x; i < n_acc; i += kHistThreads) h[i] = 0;
    }
    __syncthreads();
    const int fl = threadIdx.x & (FG - 1);
    const int slot = threadIdx.x >> fg_shift;
    const int n_slots = kHistThreads >> fg_shift;
    const int row_stride = (D + 1) * FG;
    const int fslot = g * FG + fl;
    const uint16_t *cbase = codes + (static_cast<size_t>(fslot >> 4) * n_rows) * kCodeGroup + (fslot & (kCodeGroup - 1));
    const int32_t *rlist = rows + ck.start;
    const int qlane = fl < (DT ? DT : 1) ? fl : (DT ? DT - 1 : 0);   // lanes >= D re-load the last value (never used)
    // DT path: block-uniform base pointers (SGPR pair) + 32-bit per-lane byte offsets, so that every load carries ONE address
    // VGPR to the memory pipeline instead of two (`global_load v, v_off, s[base]`).  The VGPR -> LDS / TA transfer path is shared
    // with the LDS atomics' address + data (2 clk per source dword per wave-instruction): with 64-bit VGPR addresses the two
    // gathered loads per row cost 8 clk of that path per wave-iteration on top of the 40 clk of the 9 atomics (52.5 measured);
    // the launcher takes this kernel only when n_rows * 64 fits 32 bits.
    const char *cgroup = reinterpret_cast<const char *>(codes + static_cast<size_t>(g) * n_rows * kCodeGroup);   // FG == 16: group g
    const char *qbase = reinterpret_cast<const char *>(qg);
    const uint32_t coff = static_cast<uint32_t>(fl) * 2u, qoff = static_cast<uint32_t>(qlane) * 4u;
    auto ld_code = [&](int row) -> int { return Ld::code(cgroup, static_cast<uint32_t>(row), coff); };
    auto ld_q = [&](int row) -> int { return Ld::template grad<DT>(qbase, static_cast<uint32_t>(row), qoff); };
    int p0 = slot;
    if (DT && PIPE && ck.len > 0) {
        // Software-pipelined main loop (three stages, U rows per slot and stage): while the 9U atomics of iteration i occupy the
        // LDS atomic unit, the codes / gradients of iteration i+1 (their row ids were loaded one iteration earlier) and the row
        // ids of iteration i+2 are already in flight.  Without it every wave of the block (one block per CU) waits for the two
        // dependent global-memory round trips at the same time and the atomic unit idles for a third of the kernel.
        // Prefetch positions beyond the chunk are clamped to its last row (loaded, never accumulated).
        const int last = ck.len - 1, step = n_slots * U;
        int rowB[U], codeA[U], qA[U];
        {
            int rowA[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rowA[u] = rlist[min(p0 + u * n_slots, last)];
#pragma unroll
            for (int u = 0; u < U; ++u) rowB[u] = rlist[min(p0 + step + u * n_slots, last)];
#pragma unroll
            for (int u = 0; u < U; ++u) codeA[u] = ld_code(rowA[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) qA[u] = ld_q(rowA[u]);
        }
        for (; p0 + (U - 1) * n_slots < ck.len; p0 += step) {
            int codeB[U], qB[U], rowC[U];
#pragma unroll
            for (int u = 0; u < U; ++u) codeB[u] = ld_code(rowB[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) qB[u] = ld_q(rowB[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) rowC[u] = rlist[min(p0 + 2 * step + u * n_slots, last)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int32_t *dst = h + codeA[u] * row_stride + fl;
                RowAtomics<DT, 0>::run(dst, 16, qA[u]);
                if (COUNT) atomicAdd(dst + DT * 16, 1);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { codeA[u] = codeB[u]; qA[u] = qB[u]; rowB[u] = rowC[u]; }
        }
    }
    // main loop: U rows per slot, no bounds checks inside => straight-line code: 3U independent loads, then 9U atomics
    for (; p0 + (U - 1) * n_slots < ck.len; p0 += n_slots * U) {
        int row[U], code[U];
#pragma unroll
        for (int u = 0; u < U; ++u) row[u] = rlist[p0 + u * n_slots];
        if (DT) {
#pragma unroll
            for (int u = 0; u < U; ++u) code[u] = ld_code(row[u]);
            int myq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) myq[u] = ld_q(row[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int32_t *dst = h + code[u] * row_stride + fl;
                RowAtomics<DT, 0>::run(dst, 16, myq[u]);
                if (COUNT) atomicAdd(dst + DT * 16, 1);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) code[u] = cbase[static_cast<size_t>(row[u]) * kCodeGroup];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t *q = qg + static_cast<size_t>(row[u]) * D;
                int32_t *dst = h + code[u] * row_stride + fl;
                for (int d = 0; d < D; ++d) atomicAdd(dst + d * FG, q[d]);
                atomicAdd(dst + D * FG, 1);
            }
        }
    }
    // tail: one row per slot (p0 is uniform across the 16 lanes of a DPP row, so whole rows take the branch)
    for (; p0 < ck.len; p0 += n_slots) {
        const int row = rlist[p0];
        if (DT) {
            int32_t *dst = h + ld_code(row) * row_stride + fl;
            const int myq = ld_q(row);
            RowAtomics<DT, 0>::run(dst, 16, myq);
            if (COUNT) atomicAdd(dst + DT * 16, 1);
        } else {
            const int code = cbase[static_cast<size_t>(row) * kCodeGroup];
            int32_t *dst = h + code * row_stride + fl;
            const int32_t *q = qg + static_cast<size_t>(row) * D;
            for (int d = 0; d < D; ++d) atomicAdd(dst + d * FG, q[d]);
            atomicAdd(dst + D * FG, 1);
        }
    }
    __syncthreads();
    if (direct.hist) {
        // the node's only chunk: the tile is the node's histogram of this feature group (LDS index = element * FG + feature).
        // FG == 16: 256 threads take 16 elements x 16 features with the ELEMENT as the lane-fastest index, so every 16 lanes store
        // 128 consecutive bytes of one feature (a 4-way LDS bank conflict on the read side, which is the cheaper side).
        const int hslot = direct.slot_map ? direct.slot_map[ck.slot] : ck.slot;
        int64_t *dst = direct.hist + (static_cast<size_t>(hslot) * direct.Fp + static_cast<size_t>(g) * FG) * NB * (D + 1);
        const size_t fstride = static_cast<size_t>(NB) * (D + 1);
        if (FG == 16) {
            const int n_el = NB * (D + 1);
            for (int i = threadIdx.x; i < ((n_el + 15) & ~15) * 16; i += kHistThreads) {
                const int e = (i >> 8) * 16 + (i & 15), f = (i >> 4) & 15;
                if (e < n_el) dst[f * fstride + e] = h[e * 16 + f];
            }
        } else {
            for (int i = threadIdx.x; i < n_acc; i += kHistThreads) dst[(i & (FG - 1)) * fstride + (i >> fg_shift)] = h[i];
        }
        return;
    }
    int32_t *out = partials + (static_cast<size_t>(chunk_id) * n_groups + g) * n_acc;
    if ((n_acc & 3) == 0) {   // the block's 148 KB of partial sums leave in 16-byte pieces
        const int4 *h4 = reinterpret_cast<const int4 *>(h);
        int4 *o4 = reinterpret_cast<int4 *>(out);
        for (int i = threadIdx.x; i < n_acc / 4; i += kHistThreads) o4[i] = h4[i];
    } else {
        for (int i = threadIdx.x; i < n_acc; i += kHistThreads) out[i] = h[i];
    }
}

// k_hist_build_wide<P, H, U>: output dimensions beyond 16, or class counts that leave room for only 8 or 4 features per block
// (FG = 16 / P).  A 16-lane DPP row still works on ONE data row, but on FG features x P parts of the D + 1 fields: lane
// (part p, feature f) adds the fields p*H .. p*H + H - 1 (H = ceil((D + 1) / P) <= 16) of feature f.  Lane l of the row loads the
// P values (field p*H + l, p = 0..P-1; the count's constant 1 sits at field D) once, and step k hands every part ITS lane-k value
// with one row_newbcast move per part, written under that part's bank mask (a DPP bank = 4 lanes) -- P moves per atomic instead of
// D + 1 loads per lane, which is what made the run-time-D path vector-memory bound (3.1 ms per tree at D = 18 against 1.5 ms of
// atomic-unit time).  Same LDS layout [class][field][FG] and the same partials as the generic path.
template <int P, int H, int K, int SRC>
struct WideValue {   // value of step K for this lane's part, assembled from the row's lane K
    static __device__ __forceinline__ int get(const int (&r)[P]) {
        constexpr int bank_mask = P == 2 ? (SRC == 0 ? 0x3 : 0xC) : (1 << SRC);
        const int lower = WideValue<P, H, K, SRC - 1>::get(r);
        return __builtin_amdgcn_update_dpp(lower, r[SRC], 0x150 + K, 0xf, bank_mask, false);
    }
};
template <int P, int H, int K>
struct WideValue<P, H, K, 0> {
    static __device__ __forceinline__ int get(const int (&r)[P]) { return __builtin_amdgcn_mov_dpp(r[0], 0x150 + K, 0xf, 0xf, true); }
};
template <int P, int H, int K>
struct WideAtomics {
    // n_last (uniform): fields owned by the LAST part; every other part owns H.  Steps below n_last need no per-lane test.
    static __device__ __forceinline__ void run(int32_t *dst, int FG, const int (&r)[P], int n_last, bool last_part) {
        const int v = WideValue<P, H, K, P - 1>::get(r);
        if (K < n_last) atomicAdd(dst + K * FG, v);
        else if (!last_part) atomicAdd(dst + K * FG, v);
        WideAtomics<P, H, K + 1>::run(dst, FG, r, n_last, last_part);
    }
};
template <int P, int H>
struct WideAtomics<P, H, H> {
    static __device__ __forceinline__ void run(int32_t *, int, const int (&)[P], int, bool) {}
};

template <int P, int H, int U>
__global__ __launch_bounds__(kHistThreads) void k_hist_build_wide(const uint16_t *__restrict__ codes, int n_rows,
                                                                   const int32_t *__restrict__ qg, int D,
                                                                   const int32_t *__restrict__ rows, const Chunk *__restrict__ chunks,
                                                                   int n_chunks, int n_groups, int NB, int32_t *__restrict__ partials) {
    extern __shared__ int32_t h[];
    constexpr int FG = 16 / P;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int g = jj % n_groups;
    const int chunk_id = (jj / n_groups) * 8 + xcd;
    if (chunk_id >= n_chunks) return;
    const Chunk ck = chunks[chunk_id];
    if (ck.len <= 0) return;   // unused entry of a device-planned chunk table
    const int n_acc = NB * (D + 1) * FG;
    for (int i = threadIdx.x; i < n_acc; i += kHistThreads) h[i] = 0;
    __syncthreads();
    const int fl = threadIdx.x & 15;
    const int part = fl / FG, f = fl & (FG - 1);
    const int slot = threadIdx.x >> 4;
    constexpr int n_slots = kHistThreads >> 4;
    const int row_stride = (D + 1) * FG;
    const int fslot = g * FG + f;
    const uint16_t *cbase = codes + (static_cast<size_t>(fslot >> 4) * n_rows) * kCodeGroup + (fslot & (kCodeGroup - 1));
    const int32_t *rlist = rows + ck.start;
    const int n_last = D + 1 - (P - 1) * H;          // fields the last part owns (1..H); the others own H
    const bool last_part = part == P - 1;
    const int dst_off = f + part * H * FG;
    // what this lane contributes as a SOURCE: field p*H + fl of the row, for every part p
    auto load_fields = [&](int row, int (&r)[P]) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int e = p * H + fl;
            r[p] = (fl < H && e < D) ? qg[static_cast<size_t>(row) * D + e] : (e == D ? 1 : 0);
        }
    };
    int p0 = slot;
    for (; p0 + (U - 1) * n_slots < ck.len; p0 += n_slots * U) {
        int row[U], code[U], r[U][P];
#pragma unroll
        for (int u = 0; u < U; ++u) row[u] = rlist[p0 + u * n_slots];
#pragma unroll
        for (int u = 0; u < U; ++u) code[u] = cbase[static_cast<size_t>(row[u]) * kCodeGroup];
#pragma unroll
        for (int u = 0; u < U; ++u) load_fields(row[u], r[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) WideAtomics<P, H, 0>::run(h + code[u] * row_stride + dst_off, FG, r[u], n_last, last_part);
    }
    for (; p0 < ck.len; p0 += n_slots) {
        const int row = rlist[p0];
        const int code = cbase[static_cast<size_t>(row) * kCodeGroup];
        int r[P];
        load_fields(row, r);
        WideAtomics<P, H, 0>::run(h + code * row_stride + dst_off, FG, r, n_last, last_part);
    }
    __syncthreads();
    int32_t *out = partials + (static_cast<size_t>(chunk_id) * n_groups + g) * n_acc;
    for (int i = threadIdx.x; i < n_acc; i += kHistThreads) out[i] = h[i];
}

// k_hist_build_quad<R, U>: 4 features per block (FG = 4: many outputs or many classes).  A DPP quad (4 lanes) works on ONE data row:
// lane j owns feature j and loads the fields j, j + 4, j + 8, ... (R = ceil((D + 1) / 4) registers, 16 contiguous bytes per quad and
// register); field e is then handed to the quad from lane e % 4, register e / 4, with ONE quad_perm broadcast per atomic.  (Splitting the
// fields over a 16-lane row as k_hist_build_wide does would need four DPP moves per atomic here, which makes the kernel VALU-bound.)
template <int R, int E>
struct QuadAtomics {   // fields E .. 4R-1
    static __device__ __forceinline__ void run(int32_t *dst, const int (&reg)[R], int D) {
        if (E <= D) {   // uniform
            constexpr int j = E & 3;
            const int v = __builtin_amdgcn_mov_dpp(reg[E >> 2], j * 0x55, 0xf, 0xf, true);   // quad_perm:[j,j,j,j]
            atomicAdd(dst + E * 4, v);
        }
        QuadAtomics<R, E + 1>::run(dst, reg, D);
    }
};
template <int R>
struct QuadAtomics<R, 4 * R> {
    static __device__ __forceinline__ void run(int32_t *, const int (&)[R], int) {}
};

template <int R, int U>
__global__ __launch_bounds__(kHistThreads) void k_hist_build_quad(const uint16_t *__restrict__ codes, int n_rows,
                                                                   const int32_t *__restrict__ qg, int D,
                                                                   const int32_t *__restrict__ rows, const Chunk *__restrict__ chunks,
                                                                   int n_chunks, int n_groups, int NB, int32_t *__restrict__ partials) {
    extern __shared__ int32_t h[];
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int g = jj % n_groups;
    const int chunk_id = (jj / n_groups) * 8 + xcd;
    if (chunk_id >= n_chunks) return;
    const Chunk ck = chunks[chunk_id];
    if (ck.len <= 0) return;   // unused entry of a device-planned chunk table
    const int n_acc = NB * (D + 1) * 4;
    for (int i = threadIdx.x; i < n_acc; i += kHistThreads) h[i] = 0;
    __syncthreads();
    const int f = threadIdx.x & 3;
    const int slot = threadIdx.x >> 2;
    constexpr int n_slots = kHistThreads >> 2;
    const int row_stride = (D + 1) * 4;
    const int fslot = g * 4 + f;
    const uint16_t *cbase = codes + (static_cast<size_t>(fslot >> 4) * n_rows) * kCodeGroup + (fslot & (kCodeGroup - 1));
    const int32_t *rlist = rows + ck.start;
    auto load_fields = [&](int row, int (&reg)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 4 + f;
            reg[r] = e < D ? qg[static_cast<size_t>(row) * D + e] : (e == D ? 1 : 0);
        }
    };
    int p0 = slot;
    for (; p0 + (U - 1) * n_slots < ck.len; p0 += n_slots * U) {
        int row[U], code[U], reg[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) row[u] = rlist[p0 + u * n_slots];
#pragma unroll
        for (int u = 0; u < U; ++u) code[u] = cbase[static_cast<size_t>(row[u]) * kCodeGroup];
#pragma unroll
        for (int u = 0; u < U; ++u) load_fields(row[u], reg[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) QuadAtomics<R, 0>::run(h + code[u] * row_stride + f, reg[u], D);
    }
    for (; p0 < ck.len; p0 += n_slots) {
        const int row = rlist[p0];
        const int code = cbase[static_cast<size_t>(row) * kCodeGroup];
        int reg[R];
        load_fields(row, reg);
        QuadAtomics<R, 0>::run(h + code * row_stride + f, reg, D);
    }
    __syncthreads();
    int32_t *out = partials + (static_cast<size_t>(chunk_id) * n_groups + g) * n_acc;
    for (int i = threadIdx.x; i < n_acc; i += kHistThreads) out[i] = h[i];
}

// Sum the int32 chunk partials of every slot (node) into int64, reordering to hist[slot][feature][class][D+1].
// kReduceLanes chunk lanes per element: with many chunks per node (few feature groups => up to 256 chunks) a node's chunks are summed
// by 4 threads with four loads in flight each and combined through LDS; with the usual <= 32 chunks one thread per element is faster.
template <int kReduceLanes>
__global__ __launch_bounds__(256 * kReduceLanes) void k_hist_reduce(const int32_t *__restrict__ partials,
                                                                     const int32_t *__restrict__ slot_chunk_begin,
                                                                     const int32_t *__restrict__ slot_map, int n_groups, int FG, int NB,
                                                                     int D, int Fp, int64_t *__restrict__ hist, int scatter_fs,
                                                                     const uint32_t *__restrict__ root_le, int root_F, int root_B, long long root_n) {
    __shared__ int64_t part[kReduceLanes > 1 ? kReduceLanes - 1 : 1][256];
    const int n_acc = NB * (D + 1) * FG;
    const int k = blockIdx.z, g = blockIdx.y;
    const int slot = slot_map ? slot_map[k] : k;
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.y;
    const int c0 = slot_chunk_begin[k], c1 = slot_chunk_begin[k + 1];
    int64_t s = 0;
    if (i < n_acc) {
        const int32_t *src = partials + static_cast<size_t>(g) * n_acc + i;
        const size_t stride = static_cast<size_t>(n_groups) * n_acc;
        int c = c0 + lane;
        for (; c + 7 * kReduceLanes < c1; c += 8 * kReduceLanes) {   // eight independent loads in flight (integer sums: any order)
            int32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(c + u * kReduceLanes) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c + 3 * kReduceLanes < c1; c += 4 * kReduceLanes) {
            const int32_t a = src[c * stride], b = src[(c + kReduceLanes) * stride], e = src[(c + 2 * kReduceLanes) * stride],
                          f = src[(c + 3 * kReduceLanes) * stride];
            s += static_cast<int64_t>(a) + b + e + f;
        }
        for (; c < c1; c += kReduceLanes) s += src[c * stride];
    }
    if (kReduceLanes > 1) {
        if (lane > 0) part[lane - 1][threadIdx.x] = s;
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < kReduceLanes - 1; ++q) s += part[q][threadIdx.x];
        }
    }
    const bool writer = lane == 0;     // the other chunk lanes stay until the last barrier and store nothing
    // The partials have the feature as the fastest index (the LDS layout of k_hist_build), the histograms the (class, field) pair:
    // the block's 256 sums are 16 features x 16 consecutive (class, field) elements when FG == 16.  Through a small LDS tile every
    // 16 lanes write 128 consecutive bytes of one feature instead of 16 features x 8 bytes.
    if (FG == 16) {
        __shared__ int64_t tile[16][17];
        if (writer) tile[threadIdx.x & 15][threadIdx.x >> 4] = s;
        __syncthreads();
        if (!writer) return;
        const int fl = threadIdx.x >> 4, el = threadIdx.x & 15;
        const int e = blockIdx.x * 16 + el;                       // (class, field) element: cls * (D + 1) + d
        const int f = g * FG + fl;
        if (e >= NB * (D + 1)) return;
        int64_t v = tile[fl][el];
        if (root_le && f < root_F) {   // the root's class counts from the selection: #{keys <= thr[c]} - #{keys <= thr[c-1]} (k_hist_build<COUNT = false>)
            const int cls = e / (D + 1);
            if (e - cls * (D + 1) == D) {
                const uint32_t *le = root_le + static_cast<size_t>(f) * root_B;
                const long long hi = cls < root_B ? static_cast<long long>(le[cls]) : root_n;
                const long long lo = cls == 0 ? 0ll : (cls - 1 < root_B ? static_cast<long long>(le[cls - 1]) : root_n);
                v = cls <= root_B ? hi - lo : 0;
            }
        }
        if (scatter_fs > 0) {   // send layout of the feature reduce-scatter: [owner rank][node k][feature inside the slice][class][D+1]
            if (f < Fp) hist[((static_cast<size_t>(f / scatter_fs) * gridDim.z + k) * scatter_fs + f % scatter_fs) * NB * (D + 1) + e] = v;
            return;
        }
        hist[(static_cast<size_t>(slot) * Fp + f) * NB * (D + 1) + e] = v;
        return;
    }
    if (!writer || i >= n_acc) return;
    const int fl = i % FG, d = (i / FG) % (D + 1), cls = i / (FG * (D + 1));
    const int f = g * FG + fl;
    if (scatter_fs > 0) {   // send layout of the feature reduce-scatter: [owner rank][node k][feature inside the slice][class][D+1]
        if (f < Fp) hist[(((static_cast<size_t>(f / scatter_fs) * gridDim.z + k) * scatter_fs + f % scatter_fs) * NB + cls) * (D + 1) + d] = s;
        return;
    }
    hist[((static_cast<size_t>(slot) * Fp + f) * NB + cls) * (D + 1) + d] = s;
}

__global__ void k_hist_place_slice(const int64_t *__restrict__ recv, int64_t *__restrict__ hist, const int32_t *__restrict__ slot_map, int fs,
                                   int lo, int Fp, size_t feat_elems) {
    const int k = blockIdx.z, f = blockIdx.y;
    if (lo + f >= Fp) return;
    const int64_t *src = recv + (static_cast<size_t>(k) * fs + f) * feat_elems;
    int64_t *dst = hist + (static_cast<size_t>(slot_map[k]) * Fp + lo + f) * feat_elems;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < feat_elems; i += static_cast<size_t>(gridDim.x) * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_publish_block(uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int words, uint32_t *flag,
                                                       uint32_t seq, int zero_src) {
    for (int i = threadIdx.x; i < words; i += 256) {
        __builtin_nontemporal_store(src[i], &dst[i]);
        if (zero_src) src[i] = 0;      // accumulators handed back clean: the next tree needs no memset
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Two device arrays copied into pinned, device-mapped host memory by one launch (no flag: whoever reads them has polled a later
// publication of the same stream).
__global__ __launch_bounds__(256) void k_publish_pair(const uint32_t *__restrict__ a, uint32_t *__restrict__ ha, int a_words,
                                                      const uint32_t *__restrict__ b, uint32_t *__restrict__ hb, int b_words) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a_words) __builtin_nontemporal_store(a[i], &ha[i]);
    if (i < b_words) __builtin_nontemporal_store(b[i], &hb[i]);
}
__global__ __launch_bounds__(256) void k_fetch_segments(FetchSegments fs) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < fs.n && i < fs.words[k]) static_cast<uint32_t *>(fs.dst[k])[i] = static_cast<const uint32_t *>(fs.src[k])[i];
}
__global__ __launch_bounds__(256) void k_stage_copy(StageSegments ss, const char *__restrict__ stage) {
    const int k = blockIdx.y;
    if (k >= ss.n) return;
    char *dst = static_cast<char *>(ss.dst[k]);
    const char *src = stage + ss.src_off[k];
    const uint32_t n = ss.bytes[k];
    const uint32_t i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 3u) == 0) {
        for (uint32_t i = i0; i < n / 4; i += stride) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
        for (uint32_t i = (n & ~3u) + i0; i < n; i += stride) dst[i] = src[i];
    } else {
        for (uint32_t i = i0; i < n; i += stride) dst[i] = src[i];
    }
}
__global__ __launch_bounds__(256) void k_fill_segments(FillSegments fs) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < fs.n && i < fs.words[k]) static_cast<uint32_t *>(fs.dst[k])[i] = fs.value[k];
}
__global__ void k_fill_f32(float *__restrict__ p, size_t n, float v) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// see kernels.h: winner_pack / winner_adopt
__global__ void k_winner_pack(const int32_t *__restrict__ best_idx, const float *__restrict__ best_score, const int64_t *__restrict__ counts4,
                              int max_front, int n_win, int n_act, int rank, int64_t *__restrict__ gather, int P) {
    // (the other ranks' rows are cleared here: the sum over ranks is a gather; round 5: no memset launch in front of this one)
    const size_t stride = static_cast<size_t>(n_win) + 2 * n_act;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < stride * P; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        if (i / stride != static_cast<size_t>(rank)) gather[i] = 0;
    int64_t *row = gather + static_cast<size_t>(rank) * (n_win + 2 * n_act);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_win) {
        const uint64_t key = (static_cast<uint64_t>(float_to_key(best_score[t])) << 32) | (0xffffffffu - static_cast<uint32_t>(best_idx[t]));
        row[t] = static_cast<int64_t>(key);
    }
    if (t < n_act) {
        row[n_win + 2 * t] = counts4[t];                                           // total rows of node t (global histogram)
        row[n_win + 2 * t + 1] = counts4[static_cast<size_t>(max_front) + t];      // rows right of this rank's best candidate
    }
}
__global__ void k_winner_adopt(const int64_t *__restrict__ gather, int P, int n_win, int n_act, int oblivious,
                               const int32_t *__restrict__ ref_to_internal, const int32_t *__restrict__ cand_slot, const FeatureSlot *__restrict__ slots,
                               const int32_t *__restrict__ seg_start, const uint32_t *__restrict__ thr_keys, int B, int32_t *__restrict__ best_idx,
                               float *__restrict__ best_score, int64_t *__restrict__ counts4, int max_front, NodeSplit *__restrict__ out,
                               int32_t *__restrict__ cursors) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_act) return;
    const int w = oblivious ? 0 : k;
    const size_t stride = static_cast<size_t>(n_win) + 2 * n_act;
    uint64_t best_key = 0;
    int owner = 0;
    for (int r = 0; r < P; ++r) {
        const uint64_t key = static_cast<uint64_t>(gather[r * stride + w]);
        if (r == 0 || key > best_key) { best_key = key; owner = r; }
    }
    const float score = key_to_float(static_cast<uint32_t>(best_key >> 32));
    const int idx = static_cast<int>(0xffffffffu - static_cast<uint32_t>(best_key & 0xffffffffu));
    const long long tot = gather[owner * stride + n_win + 2 * k], right = gather[owner * stride + n_win + 2 * k + 1];
    if (k == w || !oblivious) { best_idx[w] = idx; best_score[w] = score; }
    counts4[k] = tot;
    counts4[static_cast<size_t>(max_front) + k] = right;
    counts4[2 * static_cast<size_t>(max_front) + k] = 0;   // this rank's rows going right: counted next (k_count_right adds into it)
    const int j = ref_to_internal[idx];
    const int fs = cand_slot[j];
    const FeatureSlot sl = slots[fs];
    NodeSplit q{};
    q.fslot = fs;
    q.bin = sl.is_cat ? (j - sl.cand_base + 1) : (j - sl.cand_base);
    q.is_cat = sl.is_cat;
    q.thr_key = (!sl.is_cat && thr_keys) ? thr_keys[static_cast<size_t>(fs) * B + q.bin] : 0u;
    q.do_split = oblivious ? (score != -INFINITY) : (score >= 0.0f);
    q.seg_start = seg_start[k];
    q.n_left = static_cast<int>(tot - right);   // global; k_localize_splits replaces it by this rank's
    cursors[2 * k] = 0;
    cursors[2 * k + 1] = 0;
    out[k] = q;
}

// ------------------------------------------------------------------------------------------------------------
// A6/A7  candidate scores from the exact histograms.  One block per (node, feature slot).
//   numeric candidate k : right = classes > k (suffix sum), left = total - right
//   categorical cand. j : right = class j+1,                left = total - right
//   L2     : |S_L|^2/n_L + |S_R|^2/n_R            (== n_L|mean_L|^2 + n_R|mean_R|^2, node.cpp:360-373)
//   Cosine : sqrt of the same quantity, 0 if it is 0 (== cosine_score, math_ops.h:538-575, since
//            sum_{i in S} g_i . mean_S = |sum_S g|^2 / n_S)
// evaluated in fp64 from integer sums and rounded once to fp32.  -inf when the candidate repeats a condition of the
// node's path (node.cpp:154-166) or a side has fewer than min_data_in_leaf rows (node.cpp:354).
// Block (slot 0) also emits the node's parent score for greedy growth (split_candidate_generator.cpp:262-320).
// ------------------------------------------------------------------------------------------------------------
constexpr int kScoreLoads = 10;
__global__ __launch_bounds__(256, 5) void k_score(int64_t *__restrict__ hist, const int64_t *__restrict__ hist_prev,
                                               const int32_t *__restrict__ sub_par, const int32_t *__restrict__ sub_sib, int Fp, int NB, int D,
                                               const FeatureSlot *__restrict__ slots, const float *__restrict__ thr,
                                               int B, int n_cand, int min_data, int cosine, const StepScales *__restrict__ scp,
                                               const int32_t *__restrict__ path_len, const int32_t *__restrict__ path_slot,
                                               const float *__restrict__ path_val, const int32_t *__restrict__ path_bin,
                                               float *__restrict__ scores, float *__restrict__ parent,
                                               const float *__restrict__ cand_w, const int32_t *__restrict__ cand_ref,
                                               const int32_t *__restrict__ is_root, float *__restrict__ part_v, int32_t *__restrict__ part_i, int slot0,
                                               int keep_derived, float *__restrict__ part_s /*nullable: second-best DISTINCT gain per block (near-tie detection)*/,
                                               int32_t *__restrict__ part_n /*with part_s: rows the block's best sends right*/,
                                               int32_t *__restrict__ cand_nr /*nullable, scores mode: [n_nodes][n_cand] rows every candidate sends right*/) {
    extern __shared__ int64_t sh64[];  // [NB][D+1] suffix sums (numeric) or raw classes (categorical)
    const double inv_scale = scp->inv_scale;
    const int node = blockIdx.y, fs = slot0 + blockIdx.x;
    const FeatureSlot sl = slots[fs];
    const int W = D + 1;
    // classes this slot really has (numeric: n_bins thresholds + 1; categorical: its candidates + "none of them"): the slices are NB
    // classes apart, but the classes beyond NBe hold nothing -- not loaded, not scanned, not written back (a 33-class categorical
    // slot beside 257-class numeric ones moved 8x the bytes it needed; k_resolve_splits applies the same bound)
    const int NBe = min(NB, sl.n_cand + 1);
    int64_t *src = hist + (static_cast<size_t>(node) * Fp + fs) * NB * W;
    // the node's path conditions on THIS feature slot, staged once per block (a candidate that repeats one of them is rejected,
    // node.cpp:154-166); reading the path arrays from global memory inside the candidate loop costs a memory round trip per
    // path entry and thread at the deep levels.  The first wave reads one path entry per lane (kMaxPath <= 64) and the four loads
    // are issued HERE, unconditionally (the arrays hold kMaxPath entries per node), so that they are in flight with the histogram
    // loads below instead of three dependent round trips after the scan; they are consumed just before the loads' barrier.
    __shared__ float s_pval[kMaxPath];
    __shared__ int s_pbin[kMaxPath];
    __shared__ int s_np;
    static_assert(kMaxPath <= kWave, "one lane per path entry");
    int q_len = 0, q_slot = -1, q_bin = 0;
    float q_val = 0.0f;
    if (threadIdx.x < kMaxPath) {
        q_len = path_len[node];
        q_slot = path_slot[node * kMaxPath + threadIdx.x];
        q_val = path_val[node * kMaxPath + threadIdx.x];
        q_bin = path_bin[node * kMaxPath + threadIdx.x];
    }
    const int par = sub_par ? sub_par[node] : -1;
    if (par >= 0) {
        // sibling subtraction fused here: this node was not accumulated from the data; its histogram is parent - sibling (exact
        // integers).  The slice is written back because the next level subtracts from it and k_resolve_splits reads it -- except at
        // the LAST level (keep_derived == 0): nothing subtracts from it any more, and k_resolve_splits derives the winner's class
        // counts from the same two slices (one third of this kernel's traffic at the deepest, most expensive level).
        const int sib = sub_sib[node];
        const int64_t *pp = hist_prev + (static_cast<size_t>(par) * Fp + fs) * NB * W;
        const int64_t *ss = sib >= 0 ? hist + (static_cast<size_t>(sib) * Fp + fs) * NB * W : nullptr;
        // kScoreLoads elements per thread in flight: the whole slice of a 257-class, 9-field slot (2313 words / 256 threads) in ONE batch --
        // a block is a chain of dependent phases and every extra batch is a memory round trip on it
        const int tot = NBe * W, step = static_cast<int>(blockDim.x);
        for (int i0 = threadIdx.x; i0 < tot; i0 += kScoreLoads * step) {
            int64_t a[kScoreLoads], b[kScoreLoads];
#pragma unroll
            for (int u = 0; u < kScoreLoads; ++u) {
                const int i = i0 + u * step;
                a[u] = i < tot ? pp[i] : 0;
                b[u] = (ss && i < tot) ? ss[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < kScoreLoads; ++u) {
                const int i = i0 + u * step;
                if (i < tot) { const int64_t v = a[u] - b[u]; sh64[i] = v; if (keep_derived) src[i] = v; }
            }
        }
    } else {
        const int tot = NBe * W, step = static_cast<int>(blockDim.x);
        for (int i0 = threadIdx.x; i0 < tot; i0 += kScoreLoads * step) {
            int64_t a[kScoreLoads];
#pragma unroll
            for (int u = 0; u < kScoreLoads; ++u) { const int i = i0 + u * step; a[u] = i < tot ? src[i] : 0; }
#pragma unroll
            for (int u = 0; u < kScoreLoads; ++u) { const int i = i0 + u * step; if (i < tot) sh64[i] = a[u]; }
        }
    }
    if (threadIdx.x < kWave) {
        const int p = threadIdx.x;
        const bool hit = p < kMaxPath && p < q_len && q_slot == fs;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int pos = __popcll(m & ((1ull << p) - 1));
            s_pval[pos] = q_val;
            s_pbin[pos] = q_bin;
        }
        if (p == 0) s_np = __popcll(m);
    }
    __syncthreads();
    // totals = sum over all classes; also turn numeric features into suffix sums in place.  Block-wide scan: thread t of a
    // 256-class tile owns class NB-1-(tile*256+t), so an inclusive prefix over t is the suffix sum over classes.  Up to 9
    // fields are scanned together so that their cross-lane shuffles overlap.
    int64_t *total = sh64 + static_cast<size_t>(NB) * W;  // [D+1]
    double *total_f = reinterpret_cast<double *>(total + W);   // [D+1] the same totals as doubles
    constexpr int WCH = 9;
    __shared__ long long wsum[4][WCH];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int w0 = 0; w0 < W; w0 += WCH) {
        long long run_base[WCH];
#pragma unroll
        for (int j = 0; j < WCH; ++j) run_base[j] = 0;
        // (a last tile of at most 4 classes -- 257 classes = 256 + 1 is the usual shape -- is added serially below instead of
        // paying a block-wide scan and two barriers for it)
        const int n_tiles = (NBe + 255) / 256 - ((NBe % 256) != 0 && (NBe % 256) <= 4 && NBe > 256 ? 1 : 0);
        for (int tile = 0; tile < n_tiles; ++tile) {
            const int c = NBe - 1 - (tile * 256 + static_cast<int>(threadIdx.x));
            long long v[WCH];
#pragma unroll
            for (int j = 0; j < WCH; ++j) v[j] = (c >= 0 && w0 + j < W) ? sh64[c * W + w0 + j] : 0;
            static_assert(WCH == 9, "wave_scan9");
            wave_scan9(v);   // within rows of 16 lanes (row_shr 1, 2, 4, 8), then across the rows (row_bcast 15, 31)
            if (lane == kWave - 1) {
#pragma unroll
                for (int j = 0; j < WCH; ++j) wsum[wave][j] = v[j];
            }
            __syncthreads();
            {   // wave by wave (not all 36 partial sums at once: they would be the kernel's register peak and cost a wave of occupancy)
                long long below[WCH], all[WCH];
#pragma unroll
                for (int j = 0; j < WCH; ++j) { below[j] = 0; all[j] = 0; }
#pragma unroll 1
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < WCH; ++j) { const long long x = wsum[i][j]; if (i < wave) below[j] += x; all[j] += x; }
                }
#pragma unroll
                for (int j = 0; j < WCH; ++j) {
                    if (c >= 0 && !sl.is_cat && w0 + j < W) sh64[c * W + w0 + j] = v[j] + run_base[j] + below[j];
                    run_base[j] += all[j];
                }
            }
            __syncthreads();
        }
        const int rem = NBe - n_tiles * 256;       // classes 0 .. rem-1 not covered by a tile (0 or 1..4): suffix sums by one thread per field
        if (static_cast<int>(threadIdx.x) < WCH && w0 + static_cast<int>(threadIdx.x) < W) {
            const int j = threadIdx.x;
            long long run = 0;
#pragma unroll
            for (int jj = 0; jj < WCH; ++jj) if (jj == j) run = run_base[jj];
            for (int c = rem - 1; c >= 0; --c) {
                run += sh64[c * W + w0 + j];
                if (!sl.is_cat) sh64[c * W + w0 + j] = run;
            }
            total[w0 + j] = run;
            total_f[w0 + j] = static_cast<double>(run);   // exact (|sums| < 2^53): the left side below is total_f - (double)right, no int64 subtraction + second conversion
        }
    }
    __syncthreads();
    const int64_t n_tot = total[D];
    // parent score of the node (greedy growth): the totals are the same integers for every feature of the node, so every block
    // derives the identical float
    float par_score = 0.0f;
    if (blockIdx.x == 0 || part_v) {   // oblivious growth needs it from one block only
        const double x = side_term(total, D, n_tot, inv_scale);
        par_score = static_cast<float>(cosine ? sqrt(x) : x);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) parent[node] = par_score;
    // greedy growth (part_v != null): the block also reduces its candidates to one best (gain, reference index) -- stage 1 of the
    // arg-max, fused here so that a greedy level needs no separate arg-max launch (fitter.cpp:318-354: gain = fma(score, w, -parent),
    // root parent = 0, lowest reference index among maxima)
    const float par_sub = (part_v && is_root[node]) ? 0.0f : par_score;
    Best mine{-INFINITY, 0x7fffffff};
    float second = -INFINITY;     // the best gain of a candidate of another class (score_common.h second_merge)
    int mine_nr = -1;             // rows the block's best sends right
    const int np = s_np;
    for (int k = threadIdx.x; k < sl.n_cand; k += blockDim.x) {
        const int64_t *R = sh64 + (k + 1) * W;
        const int64_t n_r = R[D], n_l = n_tot - n_r;
        bool reject = (n_l < min_data) || (n_r < min_data);
        if (np > 0) {
            const float tk = sl.is_cat ? 0.0f : thr[static_cast<size_t>(fs) * B + k];
            for (int p = 0; p < np; ++p) {
                if (sl.is_cat) reject |= (s_pbin[p] == k + 1);
                else reject |= (s_pval[p] == tk);
            }
        }
        float out;
        if (reject) {
            out = -INFINITY;
        } else {
            // (score_common.h: the fused RL-sized growth kernel evaluates the same expression.)  k_score is bound by its VALU instruction
            // count and this loop was 170 of them per candidate (now ~115).
            out = candidate_score([&](int d) { return static_cast<double>(R[d]); }, total_f, D, n_l, n_r, cosine, inv_scale);
        }
        if (part_v) {
            const int j = sl.cand_base + k;
            const float gain = fmaf(out, cand_w[j], -par_sub);
            const Best cb{gain, cand_ref[j]};
            const int cls = part_n ? near_class(n_r, n_tot) : 0;     // (part_n == nullptr: batches of <= 8192 rows, classes by the gain alone)
            second = second_merge(mine.v, mine_nr, second, gain, cls, -INFINITY);
            if (better_takes_second(mine, cb)) { mine = cb; mine_nr = cls; }
        } else {
            scores[static_cast<size_t>(node) * n_cand + sl.cand_base + k] = out;
            if (cand_nr) cand_nr[static_cast<size_t>(node) * n_cand + sl.cand_base + k] = near_class(n_r, n_tot);
        }
    }
    if (part_v) {
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const Best other{__shfl_xor(mine.v, o, kWave), __shfl_xor(mine.i, o, kWave)};
            if (part_s) {
                const int onr = __shfl_xor(mine_nr, o, kWave);
                second = second_merge(mine.v, mine_nr, second, other.v, onr, __shfl_xor(second, o, kWave));
                if (better_takes_second(mine, other)) mine_nr = onr;
            }
            mine = better(mine, other);
        }
        __shared__ float bv[4], b2[4];
        __shared__ int bi[4], bn[4];
        if (lane == 0) { bv[wave] = mine.v; bi[wave] = mine.i; b2[wave] = second; bn[wave] = mine_nr; }
        __syncthreads();
        if (threadIdx.x == 0) {
            Best b{bv[0], bi[0]};
            float s2 = b2[0];
            int nr = bn[0];
            for (int q = 1; q < 4; ++q) {
                const Best o{bv[q], bi[q]};
                s2 = second_merge(b.v, nr, s2, o.v, bn[q], b2[q]);
                if (better_takes_second(b, o)) { b = o; nr = bn[q]; }
            }
            part_v[static_cast<size_t>(node) * gridDim.x + blockIdx.x] = b.v;
            part_i[static_cast<size_t>(node) * gridDim.x + blockIdx.x] = b.i;
            if (part_s) part_s[static_cast<size_t>(node) * gridDim.x + blockIdx.x] = s2;
            if (part_n) part_n[static_cast<size_t>(node) * gridDim.x + blockIdx.x] = nr;
        }
    }
}

// Two stages: stage 1 (many blocks) reduces a slice of the candidates; the per-block bests are reduced by k_resolve_splits.
constexpr int kArgmaxThreads = 256;
__global__ __launch_bounds__(kArgmaxThreads) void k_argmax_stage1(const float *__restrict__ scores, int n_nodes, int n_cand,
                                                                  const float *__restrict__ w, const int32_t *__restrict__ ref,
                                                                  const float *__restrict__ parent, const int32_t *__restrict__ is_root,
                                                                  int oblivious, float *__restrict__ part_v, int32_t *__restrict__ part_i,
                                                                  float *__restrict__ part_s /*nullable: second-best distinct score per block*/) {
    const int j = blockIdx.x * kArgmaxThreads + threadIdx.x;
    const int node = blockIdx.y;
    Best mine{-INFINITY, 0x7fffffff};
    if (j < n_cand) {
        float sc;
        if (oblivious) {   // sum over nodes in node order, fp32, then * w (fitter.cpp:426-435)
            sc = 0.0f;
            int nd = 0;
            for (; nd + 8 <= n_nodes; nd += 8) {      // eight loads in flight, added in node order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = scores[static_cast<size_t>(nd + u) * n_cand + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) sc += v[u];
            }
            for (; nd < n_nodes; ++nd) sc += scores[static_cast<size_t>(nd) * n_cand + j];
            sc = sc * w[j];
        } else {           // fma(score, w, -parent): the reference's "score*w - parent" is contracted (fitter.cpp:332)
            const float par = is_root[node] ? 0.0f : parent[node];
            sc = fmaf(scores[static_cast<size_t>(node) * n_cand + j], w[j], -par);
        }
        mine = better(mine, Best{sc, ref[j]});
    }
    __shared__ float sv[kArgmaxThreads], s2[kArgmaxThreads];
    __shared__ int si[kArgmaxThreads];
    sv[threadIdx.x] = mine.v;
    si[threadIdx.x] = mine.i;
    s2[threadIdx.x] = -INFINITY;
    __syncthreads();
    for (int o = kArgmaxThreads / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            Best a2{sv[threadIdx.x], si[threadIdx.x]}, b2{sv[threadIdx.x + o], si[threadIdx.x + o]};
            s2[threadIdx.x] = second_distinct(a2.v, s2[threadIdx.x], b2.v, s2[threadIdx.x + o]);
            a2 = better(a2, b2);
            sv[threadIdx.x] = a2.v;
            si[threadIdx.x] = a2.i;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part_v[static_cast<size_t>(node) * gridDim.x + blockIdx.x] = sv[0];
        part_i[static_cast<size_t>(node) * gridDim.x + blockIdx.x] = si[0];
        if (part_s) part_s[static_cast<size_t>(node) * gridDim.x + blockIdx.x] = s2[0];
    }
}
// Child sizes of the selected split of every active node, straight from the histograms (one wave per node), so that the
// host learns the winner AND the sizes of its children with a single read-back.  counts4 = [total_local | right_local |
// total_global | right_global], each max_front wide.
__global__ __launch_bounds__(64) void k_resolve_splits(const float *__restrict__ part_v, const int32_t *__restrict__ part_i, int n_parts,
                                                       int32_t *__restrict__ best_idx, float *__restrict__ best_score, int oblivious,
                                                       const int32_t *__restrict__ ref_to_internal, const int32_t *__restrict__ cand_slot,
                                                       const FeatureSlot *__restrict__ slots, const int64_t *__restrict__ hist_local,
                                                       const int64_t *__restrict__ hist_global, int Fp, int NB, int D,
                                                       NodeSplit *__restrict__ out, int64_t *__restrict__ counts4, int max_front,
                                                       const int32_t *__restrict__ seg_start /*nullable*/, int32_t *__restrict__ cursors,
                                                       const uint32_t *__restrict__ thr_keys, int B, char *pub, uint32_t *pub_flag,
                                                       uint32_t pub_seq, unsigned *pub_done, const int64_t *__restrict__ hist_prev,
                                                       const int32_t *__restrict__ sub_par, const int32_t *__restrict__ sub_sib,
                                                       const float *__restrict__ part_s, const int32_t *__restrict__ part_n /*nullable: oblivious levels*/, float near_rel, const float *__restrict__ parent,
                                                       const int32_t *__restrict__ is_root, int cosine_score, long long near_rows) {
    const int node = blockIdx.x;
    // pub != nullptr: the result block [best_idx | best_score | counts4] is mirrored into pinned, device-mapped host memory of the same
    // layout and the LAST block to finish stores pub_seq to pub_flag (system scope) -- the host polls it (no publishing launch)
    int32_t *pub_idx = reinterpret_cast<int32_t *>(pub);
    float *pub_score = reinterpret_cast<float *>(pub + 4 * static_cast<size_t>(max_front));
    int64_t *pub_counts = reinterpret_cast<int64_t *>(pub + 8 * static_cast<size_t>(max_front));
    // final stage of the argmax (same total order as stage 1: higher score, then lower reference index): every block
    // reduces the per-block bests of its node (oblivious: of the level); the owner block publishes them for the host
    const int src_node = oblivious ? 0 : node;
    Best mine{-INFINITY, 0x7fffffff};
    float second = -INFINITY;
    int mine_nr = 0;          // (oblivious levels carry no child sizes: every candidate's is 0 and only the gains tell classes apart)
    for (int q = threadIdx.x; q < n_parts; q += kWave) {
        const Best other{part_v[static_cast<size_t>(src_node) * n_parts + q], part_i[static_cast<size_t>(src_node) * n_parts + q]};
        if (part_s) {
            const int onr = part_n ? part_n[static_cast<size_t>(src_node) * n_parts + q] : 0;
            second = second_merge(mine.v, mine_nr, second, other.v, onr, part_s[static_cast<size_t>(src_node) * n_parts + q]);
            if (better_takes_second(mine, other)) mine_nr = onr;
        }
        mine = better(mine, other);
    }
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const Best other{__shfl_xor(mine.v, o, kWave), __shfl_xor(mine.i, o, kWave)};
        if (part_s) {
            const int onr = __shfl_xor(mine_nr, o, kWave);
            second = second_merge(mine.v, mine_nr, second, other.v, onr, __shfl_xor(second, o, kWave));
            if (better_takes_second(mine, other)) mine_nr = onr;
        }
        mine = better(mine, other);
    }
    const int best = mine.i == 0x7fffffff ? 0 : mine.i;
    const float best_v = mine.v;
    // near-tie flag (one GPU, counts4's third array is free there): the runner-up -- the best DISTINCT gain -- is within near_rel of the
    // winner, relative to the scores' magnitude, or (greedy) the winning gain is that close to zero, where "split" and "leaf" part
    // (fitter.cpp:357).  The host then has the few candidates in the window re-scored in the reference's float32 order (neartie.hip).
    if (threadIdx.x == 0 && node == src_node && !hist_global) {
        counts4[3 * static_cast<size_t>(max_front) + node] = __float_as_int(second);   // (diagnostics: GBRL_HIP_NEARTIE_DEBUG prints it)
        if (pub) pub_counts[3 * static_cast<size_t>(max_front) + node] = __float_as_int(second);
    }
    if (threadIdx.x == 0 && node == src_node) {
        best_idx[node] = best; best_score[node] = best_v;
        if (pub) { pub_idx[node] = best; pub_score[node] = best_v; }
    }
    const int j = ref_to_internal[best];
    const int fs = cand_slot[j];
    const FeatureSlot sl = slots[fs];
    const int bin = sl.is_cat ? (j - sl.cand_base + 1) : (j - sl.cand_base);
    const int W = D + 1;
    const int NBe = min(NB, sl.n_cand + 1);   // classes of the winner's slot (k_score wrote / derived only those)
    int n_left = 0;
    for (int pass = 0; pass < (hist_global ? 2 : 1); ++pass) {
        const int64_t *src = (pass ? hist_global : hist_local) + (static_cast<size_t>(node) * Fp + fs) * NB * W;
        // sub_par != nullptr (last level, one GPU): k_score did not write the derived slices back; a derived node's counts are
        // parent - sibling here as well (sub_sib < 0: the sibling has no rows)
        const int par = (sub_par && pass == 0) ? sub_par[node] : -1;
        const int64_t *sub = nullptr;
        if (par >= 0) {
            const int sib = sub_sib[node];
            src = hist_prev + (static_cast<size_t>(par) * Fp + fs) * NB * W;
            sub = sib >= 0 ? hist_local + (static_cast<size_t>(sib) * Fp + fs) * NB * W : nullptr;
        }
        long long tot = 0, right = 0;
        for (int c0 = threadIdx.x; c0 < NBe; c0 += 8 * kWave) {     // the class counts of the winner's slice: eight loads in flight per lane
            long long n[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int c = c0 + u * kWave; n[u] = c < NBe ? src[c * W + D] : 0; }
            if (sub) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int c = c0 + u * kWave; n[u] -= c < NBe ? sub[c * W + D] : 0; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * kWave;
                tot += n[u];
                if (c < NBe && (sl.is_cat ? (c == bin) : (c > bin))) right += n[u];
            }
        }
        for (int o = kWave / 2; o > 0; o >>= 1) { tot += __shfl_xor(tot, o, kWave); right += __shfl_xor(right, o, kWave); }
        if (threadIdx.x == 0) {
            counts4[(2 * pass + 0) * static_cast<size_t>(max_front) + node] = tot;
            counts4[(2 * pass + 1) * static_cast<size_t>(max_front) + node] = right;
            if (pub) {
                pub_counts[(2 * pass + 0) * static_cast<size_t>(max_front) + node] = tot;
                pub_counts[(2 * pass + 1) * static_cast<size_t>(max_front) + node] = right;
            }
            if (pass == 0) n_left = static_cast<int>(tot - right);
            if (pass == 0 && !hist_global) {
                // near-tie flag (one GPU: counts4's third array is free there): the runner-up -- the best DISTINCT gain -- is within the window of
                // the winner, relative to the scores' magnitude, or (greedy) the winning gain is that close to zero, where "split" and "leaf"
                // part (fitter.cpp:357).  The host then has the candidates in the window re-scored in the reference's float32 order (neartie.hip).
                // A zero gain needs no replay under L2 when the winner sends every row to ONE side: the reference's split score is then its
                // parent score operation for operation (node.cpp:321-376 against split_candidate_generator.cpp:293-320) and its gain is exactly
                // 0 as well.  (Cosine divides by sqrtf in one and by a double sqrt in the other: there the last bit decides, and is replayed.)
                long long near = 0;
                if (part_s && best_v != -INFINITY && node == src_node) {   // (an oblivious level carries ONE flag; the other nodes' words are cleared:
                    float mag;                                               //  row-sharded levels count this rank's right-going rows into them next)
                    if (oblivious) mag = fabsf(best_v);
                    else { const float par = is_root[node] ? 0.0f : parent[node]; mag = fmaxf(fabsf(best_v + par), fabsf(par)); }
                    const float win = near_window_rel(near_rel, oblivious ? near_rows : tot) * mag;
                    if (second != -INFINITY && best_v - second <= win) near = 1;
                    if (!oblivious && !is_root[node] && fabsf(best_v) <= win && !(cosine_score == 0 && (right == 0 || right == tot))) near = 1;
                }
                counts4[2 * static_cast<size_t>(max_front) + node] = near;
                if (pub) pub_counts[2 * static_cast<size_t>(max_front) + node] = near;
            }
        }
    }
    if (threadIdx.x == 0) {
        NodeSplit q{};
        q.fslot = fs; q.bin = bin; q.is_cat = sl.is_cat;
        q.thr_key = (!sl.is_cat && thr_keys) ? thr_keys[static_cast<size_t>(fs) * B + bin] : 0u;
        if (seg_start) {
            // complete descriptor: the partition of this level is enqueued without waiting for the host's read-back.
            // Same decision rule as the host (fitter.cpp:357 greedy: score >= 0; fitter.cpp:458 oblivious: any finite best)
            const float bs = best_v;
            q.do_split = oblivious ? (bs != -INFINITY) : (bs >= 0.0f);
            q.seg_start = seg_start[node];
            q.n_left = n_left;
            cursors[2 * node] = 0;
            cursors[2 * node + 1] = 0;
        }
        out[node] = q;
        if (pub) {
            __threadfence_system();
            if (atomicAdd(pub_done, 1u) == gridDim.x - 1) {
                *pub_done = 0;     // ready for the next launch (launches on one stream do not overlap)
                __hip_atomic_store(pub_flag, pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// A9  partition: every splitting node's segment [seg_start, seg_start+n) of the row list is split into [left | right] in
// place of the same range of the output list.  Destination slots are handed
// out with one wave-aggregated atomic per side; the order inside a child is not the reference's stable order, which is
// harmless because everything computed downstream is an order-independent integer sum.
// ------------------------------------------------------------------------------------------------------------
constexpr int kPartThreads = 1024;
constexpr int kPartRows = 4096;  // rows per block: 4 per thread, all loads issued before the ballots
__global__ __launch_bounds__(kPartThreads) void k_partition(const int32_t *__restrict__ rows_in, int32_t *__restrict__ rows_out,
                                                            const uint16_t *__restrict__ codes, const uint32_t *__restrict__ kt, int n_rows,
                                                            const Chunk *__restrict__ chunks, const NodeSplit *__restrict__ splits,
                                                            int32_t *__restrict__ cursors) {
    const Chunk ck = chunks[blockIdx.x];
    if (ck.len <= 0) return;    // unused entry of a device-planned chunk table
    const NodeSplit sp = splits[ck.slot];
    if (!sp.do_split) return;   // the node became a leaf (its segment stays in the input list)
    const int lane = threadIdx.x & (kWave - 1);
    constexpr int U = kPartRows / kPartThreads;
    int row[U];
    bool right[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = u * kPartThreads + threadIdx.x;
        row[u] = p < ck.len ? rows_in[ck.start + p] : -1;
    }
    if (!sp.is_cat && kt) {
        // numeric split: code > bin <=> key > threshold key (the thresholds are sorted); the keys are feature-major, so the rows
        // of a node cost 4 bytes each instead of a 32-byte code record
        const uint32_t *kcol = kt + static_cast<size_t>(sp.fslot) * n_rows;
#pragma unroll
        for (int u = 0; u < U; ++u) right[u] = row[u] >= 0 && kcol[row[u]] > sp.thr_key;
    } else {
        const uint16_t *cbase = codes + (static_cast<size_t>(sp.fslot >> 4) * n_rows) * kCodeGroup + (sp.fslot & (kCodeGroup - 1));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int code = row[u] >= 0 ? cbase[static_cast<size_t>(row[u]) * kCodeGroup] : 0;
            right[u] = sp.is_cat ? (code == sp.bin) : (code > sp.bin);
        }
    }
    // destination slots: ballots give the rank inside a wave; per-(wave,u) counts are scanned in LDS and the block reserves
    // its range with ONE global atomic per side (at level 0 every wave of the grid would otherwise hit the same two words).
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
    constexpr int W = kPartThreads / kWave;
    __shared__ int cnt[W * U][2];
    __shared__ int base[2];
    const int wave = threadIdx.x / kWave;
    unsigned long long mr[U], ml[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool active = row[u] >= 0;
        mr[u] = __ballot(active && right[u]);
        ml[u] = __ballot(active && !right[u]);
        if (lane == 0) { cnt[wave * U + u][1] = __popcll(mr[u]); cnt[wave * U + u][0] = __popcll(ml[u]); }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int side = threadIdx.x;
        int run = 0;
        for (int q = 0; q < W * U; ++q) { const int c = cnt[q][side]; cnt[q][side] = run; run += c; }
        base[side] = run ? atomicAdd(&cursors[ck.slot * 2 + side], run) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (row[u] < 0) continue;
        const int off = cnt[wave * U + u][right[u] ? 1 : 0];
        const int dst = right[u] ? (sp.seg_start + sp.n_left + base[1] + off + __popcll(mr[u] & below))
                                 : (sp.seg_start + base[0] + off + __popcll(ml[u] & below));
        rows_out[dst] = row[u];
    }
}

__global__ __launch_bounds__(kPartThreads) void k_count_right(const int32_t *__restrict__ rows, const uint16_t *__restrict__ codes,
                                                              const uint32_t *__restrict__ kt, int n_rows, const Chunk *__restrict__ chunks,
                                                              const NodeSplit *__restrict__ splits, int64_t *__restrict__ n_right) {
    const Chunk ck = chunks[blockIdx.x];
    const NodeSplit sp = splits[ck.slot];
    const uint16_t *cbase = codes + (static_cast<size_t>(sp.fslot >> 4) * n_rows) * kCodeGroup + (sp.fslot & (kCodeGroup - 1));
    int c = 0;
    if (!sp.is_cat && kt) {
        const uint32_t *kcol = kt + static_cast<size_t>(sp.fslot) * n_rows;
        for (int p = threadIdx.x; p < ck.len; p += kPartThreads) c += kcol[rows[ck.start + p]] > sp.thr_key ? 1 : 0;
    } else {
        for (int p = threadIdx.x; p < ck.len; p += kPartThreads) {
            const int code = cbase[static_cast<size_t>(rows[ck.start + p]) * kCodeGroup];
            c += (sp.is_cat ? (code == sp.bin) : (code > sp.bin)) ? 1 : 0;
        }
    }
    for (int o = kWave / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
    __shared__ int wsum[kPartThreads / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tot = 0;
        for (int w = 0; w < kPartThreads / kWave; ++w) tot += wsum[w];
        if (tot) atomicAdd(reinterpret_cast<unsigned long long *>(&n_right[ck.slot]), static_cast<unsigned long long>(tot));
    }
}

// ------------------------------------------------------------------------------------------------------------
// A11  leaf sums of RAW gradients (fixed-point int64, exact) + counts.  acc[leaf][0..D) sums, acc[leaf][D] count.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_leaf_sums(const float *__restrict__ grads, int D, const int32_t *__restrict__ rows,
                                                   const Chunk *__restrict__ chunks, const StepScales *__restrict__ scp,
                                                   int64_t *__restrict__ acc) {
    extern __shared__ unsigned long long shl[];  // [D+1]
    const double scale = scp->leaf_scale;
    for (int i = threadIdx.x; i <= D; i += blockDim.x) shl[i] = 0ull;
    __syncthreads();
    const Chunk ck = chunks[blockIdx.x];
    // thread <-> (row, d): consecutive threads read consecutive d of one row
    const int per = blockDim.x / D > 0 ? blockDim.x / D : 1;
    const int d = threadIdx.x % D, sub = threadIdx.x / D;
    long long s = 0;
    if (sub < per) {
        // eight independent (row id -> gradient) load chains in flight per thread
        int p = sub;
        for (; p + 7 * per < ck.len; p += 8 * per) {
            int r[8];
            float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = rows[ck.start + p + u * per];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = grads[static_cast<size_t>(r[u]) * D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += __double2ll_rn(static_cast<double>(g[u]) * scale);
        }
        for (; p < ck.len; p += per) {
            const int row = rows[ck.start + p];
            s += __double2ll_rn(static_cast<double>(grads[static_cast<size_t>(row) * D + d]) * scale);
        }
        atomicAdd(&shl[d], static_cast<unsigned long long>(s));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += blockDim.x)
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[static_cast<size_t>(ck.slot) * (D + 1) + i]), shl[i]);
    if (threadIdx.x == 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[static_cast<size_t>(ck.slot) * (D + 1) + D]),
                  static_cast<unsigned long long>(ck.len));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- wrappers

int column_sums_blocks(int n, int D) { return grid_for(static_cast<size_t>(n) * D, 256 * 8, 1024); }

void column_sums(const float *g, int n, int D, const float *center, double *block_partials, int n_blocks, double *out,
                 hipStream_t s) {
    const int bs = D <= 256 ? (256 / D) * D : D;  // multiple of D so a thread owns one column
    hipLaunchKernelGGL(k_column_sums, dim3(n_blocks), dim3(bs), 2 * bs * sizeof(double), s, g, static_cast<size_t>(n) * D, D,
                       center, block_partials);
    hipLaunchKernelGGL(k_column_sums_final, dim3(2 * D), dim3(256), 0, s, block_partials, n_blocks, D, out);
}

// RL-sized batches: statistics + quantisation in one launch (k_small_stats).  false: shape not covered, nothing was launched.
bool small_stats(const float *g, int n, int D, bool centred, int chunk_rows, double *stat, float *meanden, StepScales *sc, int32_t *qg, hipStream_t s) {
    const int n_blocks = column_sums_blocks(n, D);
    const int bs = D <= 256 ? (256 / D) * D : D;
    if (D > 16 || n_blocks * bs > kSmallStatsVirtual || n_blocks > 32 || n < 2) return false;
    const size_t lds = sizeof(double) * (2 * static_cast<size_t>(n_blocks) * bs + static_cast<size_t>(n_blocks) * 2 * D);
    if (lds > 156 * 1024) return false;
    // A device that refuses the LDS opt-in (or the launch) is remembered: every later call answers "nothing was launched" and the
    // caller runs the seven-kernel chain (the first version reported the failure on the first call only).
    static PerDeviceOnce attr;
    static uint64_t unsupported = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0;
    if (attr.first() && hipFuncSetAttribute(reinterpret_cast<const void *>(k_small_stats), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) != hipSuccess) {
        (void)hipGetLastError();
        unsupported |= bit;
    }
    if (unsupported & bit) return false;
    hipLaunchKernelGGL(k_small_stats, dim3(1), dim3(kSmallStatsThreads), lds, s, g, n, D, n_blocks, bs, centred ? 1 : 0, chunk_rows, stat, meanden, sc, qg);
    if (hipGetLastError() != hipSuccess) { unsupported |= bit; return false; }
    return true;
}

void quantize_grads(const float *g, size_t n_el, int D, const float *mean, const float *denom, const StepScales *sc, int32_t *qg,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_quantize, dim3(grid_for(n_el, 256, 4096)), dim3(256), 0, s, g, n_el, D, mean, denom, sc, qg);
}

__global__ void k_sub_arrays(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, size_t n) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        out[i] = a[i] - b[i];
}
void sub_arrays(const float *a, const float *b, float *out, size_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_sub_arrays, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, a, b, out, n);
}
__global__ void k_gather_rows(const float *__restrict__ src, const int32_t *__restrict__ perm, float *__restrict__ dst, int n, int width) {
    const size_t total = static_cast<size_t>(n) * width;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const size_t r = i / width, c = i - r * width;
        dst[i] = src[static_cast<size_t>(perm[r]) * width + c];
    }
}
void gather_rows(const float *src, const int32_t *perm, float *dst, int n, int width, hipStream_t s) {
    const size_t total = static_cast<size_t>(n) * width;
    if (total == 0) return;
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(total, 256, 8192)), dim3(256), 0, s, src, perm, dst, n, width);
}
__global__ void k_f64_to_f32(const double *__restrict__ in, float *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = static_cast<float>(in[i]);
}
__global__ void k_f32_to_f64(const float *__restrict__ in, double *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = static_cast<double>(in[i]);
}
// Row-sharded statistics in ONE sum all-reduce: msg = [D column sums | P rows of D maxima, one per rank].  A rank writes its maxima
// into its own row and zeros into the others, so the SUM over ranks is a gather (x + 0 = x exactly; maxima are >= 0), and the maximum
// over the P rows is taken locally afterwards.  Replaces a sum all-reduce + a max all-reduce (and two conversion launches).
__global__ void k_stats_pack(const double *__restrict__ st /*[2D] sums | maxima*/, int D, int P, int rank, double *__restrict__ msg /*[D + P D + 1]*/, double extra) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) msg[D + P * D] = extra;     // one more summed word (the rank's row count in the first round)
    if (i < D) msg[i] = st[i];
    if (i < P * D) msg[D + i] = (i / D == rank) ? st[D + i % D] : 0.0;
}
__global__ void k_stats_unpack(const double *__restrict__ msg, int D, int P, double *__restrict__ st) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    st[d] = msg[d];
    double m = 0.0;
    for (int r = 0; r < P; ++r) m = fmax(m, msg[D + r * D + d]);
    st[D + d] = m;
}
void stats_pack(const double *st, int D, int P, int rank, double *msg, hipStream_t s, double extra) {
    hipLaunchKernelGGL(k_stats_pack, dim3((std::max(1, P) * D + D + 255) / 256), dim3(256), 0, s, st, D, P, rank, msg, extra);
}
void stats_unpack(const double *msg, int D, int P, double *st, hipStream_t s) {
    hipLaunchKernelGGL(k_stats_unpack, dim3((D + 255) / 256), dim3(256), 0, s, msg, D, P, st);
}
// -x for the first n floats (min over ranks = -max(-x): the column minima and maxima of uniform candidates travel in ONE max all-reduce)
__global__ void k_negate_f32(float *__restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = -p[i];
}
void negate_f32(float *p, int n, hipStream_t s) { hipLaunchKernelGGL(k_negate_f32, dim3((n + 255) / 256), dim3(256), 0, s, p, n); }
void f64_to_f32(const double *in, float *out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_f64_to_f32, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n);
}
void f32_to_f64(const float *in, double *out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_to_f64, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n);
}
void stats_mean(const double *stat, long long n, int D, float *meanden, hipStream_t s) {
    hipLaunchKernelGGL(k_stats_mean, dim3(1), dim3(256), 0, s, stat, n, D, meanden);
}
void stats_finish(const double *stat_raw, const double *stat_centred, long long n, int D, int chunk_rows, float *meanden,
                  StepScales *sc, hipStream_t s) {
    hipLaunchKernelGGL(k_stats_finish, dim3(1), dim3(256), 0, s, stat_raw, stat_centred, n, D, chunk_rows, meanden, sc);
}

void column_minmax(const uint32_t *kt, int n, int F, uint32_t *mn, uint32_t *mx, hipStream_t s) {
    const int rows_per_block = 16384;
    dim3 grid((n + rows_per_block - 1) / rows_per_block, F);
    hipLaunchKernelGGL(k_column_minmax, grid, dim3(256), 0, s, kt, n, rows_per_block, mn, mx);
}

void uniform_thresholds(const uint32_t *mn, const uint32_t *mx, int F, int B, float *thr, hipStream_t s, uint32_t *thr_keys) {
    hipLaunchKernelGGL(k_uniform_thresholds, dim3((F * B + 255) / 256), dim3(256), 0, s, mn, mx, F, B, thr, thr_keys);
}

template <bool STRICT, int FT>
static void launch_bin_rows(const float *obs, int n, int F, const uint32_t *trial_keys, int B, unsigned long long *counts, uint16_t *codes,
                            int code_stride, int code_off, hipStream_t s) {
    const int tiles = (F + FT - 1) / FT;
    const int rpi = kBinThreads / FT;
    dim3 grid(grid_for(static_cast<size_t>(n), rpi * 64, 1024), tiles);
    const size_t lds = (static_cast<size_t>(B) * FT + static_cast<size_t>(B + 1) * FT) * sizeof(uint32_t);
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_rows<STRICT, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL((k_bin_rows<STRICT, FT>), grid, dim3(kBinThreads), lds, s, obs, n, F, trial_keys, B, counts, codes, code_stride, code_off);
}

void bin_rows(const float *obs, int n, int F, const uint32_t *trial_keys, int B, bool strict, int64_t *counts_,
              uint16_t *codes, int code_stride, int code_off, hipStream_t s) {
    unsigned long long *counts = reinterpret_cast<unsigned long long *>(counts_);
    // thresholds + class counters of one feature take (2 B + 1) LDS words: 64 features per tile up to 319 thresholds, 16 up to 1279, 4 up to 5119
    auto fits = [&](int ft) { return (static_cast<size_t>(2 * B + 1) * ft) * sizeof(uint32_t) <= 160 * 1024; };
#define GBRL_BIN_ROWS(FT) do { if (strict) launch_bin_rows<true, FT>(obs, n, F, trial_keys, B, counts, codes, code_stride, code_off, s); \
                                else launch_bin_rows<false, FT>(obs, n, F, trial_keys, B, counts, codes, code_stride, code_off, s); } while (0)
    if (fits(64)) GBRL_BIN_ROWS(64);
    else if (fits(16)) GBRL_BIN_ROWS(16);
    else if (fits(4)) GBRL_BIN_ROWS(4);
    else throw std::runtime_error("the 32-pass bisection selection of quantile candidates supports n_bins <= 5119");
#undef GBRL_BIN_ROWS
}

void qsel_init(uint32_t *prefix, uint32_t *trial, int F, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_qsel_init, dim3((F * B + 255) / 256), dim3(256), 0, s, prefix, trial, F * B);
}
void qsel_update(uint32_t *prefix, uint32_t *trial, const int64_t *counts, const int64_t *cum, int F, int B, int bit,
                 int next_bit, hipStream_t s) {
    hipLaunchKernelGGL(k_qsel_update, dim3(F), dim3(256), (B + 1) * sizeof(unsigned long long), s, prefix, trial, counts,
                       cum, B, bit, next_bit);
}
void keys_to_floats(uint32_t *keys, float *out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_keys_to_floats, dim3((n + 255) / 256), dim3(256), 0, s, keys, out, n);
}
void floats_to_keys(const float *in, uint32_t *keys, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_floats_to_keys, dim3((n + 255) / 256), dim3(256), 0, s, in, keys, n);
}
void iota_rows(int32_t *rows, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, s, rows, n);
}

size_t hist_lds_bytes(int NB, int D, int FG) { return static_cast<size_t>(NB) * (D + 1) * FG * sizeof(int32_t); }

template <int DT, int U, bool PIPE, bool COUNT = true>
static void launch_hist_p(const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows, const Chunk *chunks,
                          int n_chunks, int n_groups, int FG, int shift, int NB, int32_t *partials, size_t lds, hipStream_t s,
                          hipEvent_t ev_start, hipEvent_t ev_stop, const HistDirect &direct) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist_build<DT, U, PIPE, HistLoads, COUNT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    }
    const int grid = 8 * n_groups * ((n_chunks + 7) / 8);
    // ev_start / ev_stop (nullable): the dispatch's own begin / end timestamps -- no extra packets in the stream, unlike
    // hipEventRecord around the launch
    hipExtLaunchKernelGGL((k_hist_build<DT, U, PIPE, HistLoads, COUNT>), dim3(grid), dim3(kHistThreads), lds, s, ev_start, ev_stop, 0, codes, n_rows, qg, D, rows,
                          chunks, n_chunks, n_groups, FG, shift, NB, partials, direct);
}
template <int DT, int U>
static void launch_hist(const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows, const Chunk *chunks,
                        int n_chunks, int n_groups, int FG, int shift, int NB, int32_t *partials, size_t lds, hipStream_t s,
                        hipEvent_t ev_start, hipEvent_t ev_stop, const HistDirect &direct = HistDirect{}, bool count_rows = true) {
    const bool pipe = [] { const char *e = hooks::raw(hooks::HIST_PIPE); return !(e && e[0] == '0'); }();   // measurement hook
    if constexpr (DT != 0) {
        if (pipe && !count_rows) { launch_hist_p<DT, U, true, false>(codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, FG, shift, NB, partials, lds, s, ev_start, ev_stop, direct); return; }
        if (pipe) { launch_hist_p<DT, U, true>(codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, FG, shift, NB, partials, lds, s, ev_start, ev_stop, direct); return; }
    }
    launch_hist_p<DT, U, false>(codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, FG, shift, NB, partials, lds, s, ev_start, ev_stop, direct);
}

template <int P, int H>
static void launch_hist_wide_one(const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows, const Chunk *chunks,
                                 int n_chunks, int n_groups, int NB, int32_t *partials, size_t lds, hipStream_t s, hipEvent_t ev_start,
                                 hipEvent_t ev_stop) {
    constexpr int U = 4;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist_build_wide<P, H, U>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
    }
    const int grid = 8 * n_groups * ((n_chunks + 7) / 8);
    hipExtLaunchKernelGGL((k_hist_build_wide<P, H, U>), dim3(grid), dim3(kHistThreads), lds, s, ev_start, ev_stop, 0, codes, n_rows, qg, D,
                          rows, chunks, n_chunks, n_groups, NB, partials);
}
template <int P, int H>
struct WideDispatch {
    static bool run(int h, const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows, const Chunk *chunks,
                    int n_chunks, int n_groups, int NB, int32_t *partials, size_t lds, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
        if (h == H) {
            launch_hist_wide_one<P, H>(codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, NB, partials, lds, s, ev_start, ev_stop);
            return true;
        }
        return WideDispatch<P, H - 1>::run(h, codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, NB, partials, lds, s, ev_start, ev_stop);
    }
};
template <int P>
struct WideDispatch<P, 0> {
    static bool run(int, const uint16_t *, int, const int32_t *, int, const int32_t *, const Chunk *, int, int, int, int32_t *, size_t,
                    hipStream_t, hipEvent_t, hipEvent_t) { return false; }
};
template <int R>
struct QuadDispatch {
    static bool run(int r, const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows, const Chunk *chunks,
                    int n_chunks, int n_groups, int NB, int32_t *partials, size_t lds, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
        if (r == R) {
            constexpr int U = R <= 8 ? 4 : 2;
            static PerDeviceOnce attr_set;
            if (attr_set.first()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist_build_quad<R, U>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024);
            }
            const int grid = 8 * n_groups * ((n_chunks + 7) / 8);
            hipExtLaunchKernelGGL((k_hist_build_quad<R, U>), dim3(grid), dim3(kHistThreads), lds, s, ev_start, ev_stop, 0, codes, n_rows, qg,
                                  D, rows, chunks, n_chunks, n_groups, NB, partials);
            return true;
        }
        return QuadDispatch<R - 1>::run(r, codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, NB, partials, lds, s, ev_start, ev_stop);
    }
};
template <>
struct QuadDispatch<0> {
    static bool run(int, const uint16_t *, int, const int32_t *, int, const int32_t *, const Chunk *, int, int, int, int32_t *, size_t,
                    hipStream_t, hipEvent_t, hipEvent_t) { return false; }
};

static bool launch_hist_wide(int P, int H, const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows,
                             const Chunk *chunks, int n_chunks, int n_groups, int NB, int32_t *partials, size_t lds, hipStream_t s,
                             hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (H < 1 || H > 16) return false;
    if (P == 2) return WideDispatch<2, 16>::run(H, codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, NB, partials, lds, s, ev_start, ev_stop);
    return false;
}

// (FG 8 / 4 may go to the wide / quad kernels, which only write partials -- and skip empty chunks)
bool hist_direct_supported(int FG) { return FG != 8 && FG != 4; }
// the count-less variant exists for the compile-time-D kernels with the software pipeline (what hist_build takes for FG == 16, D <= 16)
bool hist_countless_supported(int D, int FG, int n_rows) {
    const bool generic_only = [] { const char *e = hooks::raw(hooks::HIST_GENERIC); return e && e[0] == '1'; }();
    const bool pipe = [] { const char *e = hooks::raw(hooks::HIST_PIPE); return !(e && e[0] == '0'); }();
    return !generic_only && pipe && FG == 16 && D >= 1 && D <= 16 && n_rows <= (1 << 26);
}
bool hist_build(const uint16_t *codes, int n_rows, const int32_t *qg, int D, const int32_t *rows, const Chunk *chunks,
                int n_chunks, int n_groups, int FG, int NB, int32_t *partials, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop,
                const HistDirect *direct, bool count_rows) {
    const HistDirect dir = direct ? *direct : HistDirect{};
    int shift = 0;
    while ((1 << shift) < FG) ++shift;
    const size_t lds = hist_lds_bytes(NB, D, FG);
    const bool generic_only = [] { const char *e = hooks::raw(hooks::HIST_GENERIC); return e && e[0] == '1'; }();   // test hook
    // the compile-time-D kernels address codes and gradients with 32-bit byte offsets from block-uniform bases (64 bytes per row at D = 16)
    if (generic_only || n_rows > (1 << 26)) {
        launch_hist<0, 4>(codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, FG, shift, NB, partials, lds, s, ev_start, ev_stop, dir);
        return dir.hist != nullptr;
    }
#ifndef GBRL_HIST_U
#define GBRL_HIST_U 8
#endif
#define GBRL_HIST_CASE(DD) case DD: launch_hist<DD, GBRL_HIST_U>(codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, FG, shift, NB, partials, lds, s, ev_start, ev_stop, dir, count_rows); return dir.hist != nullptr;
    if (FG == 16) {
        switch (D) {
            GBRL_HIST_CASE(1) GBRL_HIST_CASE(2) GBRL_HIST_CASE(3) GBRL_HIST_CASE(4) GBRL_HIST_CASE(5) GBRL_HIST_CASE(6)
            GBRL_HIST_CASE(7) GBRL_HIST_CASE(8) GBRL_HIST_CASE(9) GBRL_HIST_CASE(10) GBRL_HIST_CASE(11) GBRL_HIST_CASE(12)
            GBRL_HIST_CASE(13) GBRL_HIST_CASE(14) GBRL_HIST_CASE(15) GBRL_HIST_CASE(16)
            default: break;
        }
    }
#undef GBRL_HIST_CASE
    // 8 features per block: the fields of a row are split over the two halves of the 16-lane DPP row (k_hist_build_wide);
    // 4 features per block: one DPP quad per data row (k_hist_build_quad)
    if (FG == 8 && D + 1 <= 32) {
        if (launch_hist_wide(2, (D + 2) / 2, codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, NB, partials, lds, s, ev_start, ev_stop)) return false;
    }
    if (FG == 4 && D + 1 <= 64) {
        if (QuadDispatch<16>::run((D + 4) / 4, codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, NB, partials, lds, s, ev_start, ev_stop)) return false;
    }
    launch_hist<0, 4>(codes, n_rows, qg, D, rows, chunks, n_chunks, n_groups, FG, shift, NB, partials, lds, s, ev_start, ev_stop, dir);
    return dir.hist != nullptr;
}

void hist_reduce(const int32_t *partials, const int32_t *slot_chunk_begin, const int32_t *slot_map, int n_slots, int n_groups, int FG,
                 int NB, int D, int Fp, int64_t *hist, hipStream_t s, int chunks_per_slot, int scatter_fs, const uint32_t *root_le, int root_F,
                 int root_B, long long root_n) {
    const int n_acc = NB * (D + 1) * FG;
    dim3 grid((n_acc + 255) / 256, n_groups, n_slots);
    if (FG != 16) root_le = nullptr;   // (the override lives in the FG == 16 store path; callers check hist_countless_supported)
    if (chunks_per_slot >= 48)
        hipLaunchKernelGGL(k_hist_reduce<4>, grid, dim3(256, 4), 0, s, partials, slot_chunk_begin, slot_map, n_groups, FG, NB, D, Fp, hist, scatter_fs, root_le, root_F, root_B, root_n);
    else
        hipLaunchKernelGGL(k_hist_reduce<1>, grid, dim3(256, 1), 0, s, partials, slot_chunk_begin, slot_map, n_groups, FG, NB, D, Fp, hist, scatter_fs, root_le, root_F, root_B, root_n);
}
void hist_place_slice(const int64_t *recv, int64_t *hist, const int32_t *slot_map, int n, int fs, int lo, int Fp, size_t feat_elems, hipStream_t s) {
    if (n <= 0 || fs <= 0) return;
    hipLaunchKernelGGL(k_hist_place_slice, dim3(static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(8, (feat_elems + 255) / 256))), fs, n), dim3(256), 0, s, recv, hist,
                       slot_map, fs, lo, Fp, feat_elems);
}
void publish_block(void *d_src, void *h_dst_mapped, size_t bytes, uint32_t *flag_mapped, uint32_t seq, hipStream_t s, bool zero_src) {
    hipLaunchKernelGGL(k_publish_block, dim3(1), dim3(256), 0, s, static_cast<uint32_t *>(d_src), static_cast<uint32_t *>(h_dst_mapped),
                       static_cast<int>(bytes / 4), flag_mapped, seq, zero_src ? 1 : 0);
}
void publish_pair(const void *d_a, void *h_a_mapped, size_t a_bytes, const void *d_b, void *h_b_mapped, size_t b_bytes, hipStream_t s) {
    const int aw = static_cast<int>(a_bytes / 4), bw = static_cast<int>(b_bytes / 4);
    hipLaunchKernelGGL(k_publish_pair, dim3((std::max(aw, bw) + 255) / 256), dim3(256), 0, s, static_cast<const uint32_t *>(d_a),
                       static_cast<uint32_t *>(h_a_mapped), aw, static_cast<const uint32_t *>(d_b), static_cast<uint32_t *>(h_b_mapped), bw);
}
void fetch_segments(const FetchSegments &fs, hipStream_t s) {
    uint32_t mx = 0;
    for (int k = 0; k < fs.n; ++k) mx = std::max(mx, fs.words[k]);
    if (mx) hipLaunchKernelGGL(k_fetch_segments, dim3((mx + 255) / 256), dim3(256), 0, s, fs);
}
void stage_copy(const StageSegments &ss, const void *stage_mapped, hipStream_t s) {
    uint32_t mx = 0;
    for (int k = 0; k < ss.n; ++k) mx = std::max(mx, ss.bytes[k]);
    if (!mx || ss.n <= 0) return;
    const unsigned bx = std::min<unsigned>(64, (mx / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_stage_copy, dim3(bx, ss.n), dim3(256), 0, s, ss, static_cast<const char *>(stage_mapped));
}
void fill_segments(const FillSegments &fs, hipStream_t s) {
    uint32_t mx = 0;
    for (int k = 0; k < fs.n; ++k) mx = std::max(mx, fs.words[k]);
    if (mx) hipLaunchKernelGGL(k_fill_segments, dim3((mx + 255) / 256), dim3(256), 0, s, fs);
}
void fill_f32(float *p, size_t n, float v, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fill_f32, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}
void winner_pack(const int32_t *best_idx, const float *best_score, const int64_t *counts4, int max_front, int n_win, int n_act, int rank,
                 int64_t *gather, hipStream_t s, int P) {
    hipLaunchKernelGGL(k_winner_pack, dim3((n_act + 63) / 64), dim3(64), 0, s, best_idx, best_score, counts4, max_front, n_win, n_act, rank, gather, P);
}
void winner_adopt(const int64_t *gather, int P, int n_win, int n_act, bool oblivious, const int32_t *ref_to_internal, const int32_t *cand_slot,
                  const FeatureSlot *slots, const int32_t *seg_start, const uint32_t *thr_keys, int B, int32_t *best_idx, float *best_score,
                  int64_t *counts4, int max_front, NodeSplit *out, int32_t *cursors, hipStream_t s) {
    hipLaunchKernelGGL(k_winner_adopt, dim3((n_act + 63) / 64), dim3(64), 0, s, gather, P, n_win, n_act, oblivious ? 1 : 0, ref_to_internal, cand_slot,
                       slots, seg_start, thr_keys, B, best_idx, best_score, counts4, max_front, out, cursors);
}

void score_candidates(int64_t *hist, const int64_t *hist_prev, const int32_t *sub_par, const int32_t *sub_sib, int n_nodes, int Fp, int NB,
                      int D, const FeatureSlot *slots, int n_slots,
                      const float *thr, int B, int n_cand, int min_data, int cosine, const StepScales *sc,
                      const int32_t *path_len, const int32_t *path_slot, const float *path_val, const int32_t *path_bin,
                      float *scores, float *parent, const float *cand_w, const int32_t *cand_ref, const int32_t *is_root, float *part_v,
                      int32_t *part_i, hipStream_t s, int slot0, bool keep_derived, float *part_s, int32_t *part_n, int32_t *cand_nr) {
    const size_t lds = static_cast<size_t>(NB + 2) * (D + 1) * sizeof(int64_t);   // class sums, the totals, the totals as doubles
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_score), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }
    hipLaunchKernelGGL(k_score, dim3(n_slots, n_nodes), dim3(256), lds, s, hist, hist_prev, sub_par, sub_sib, Fp, NB, D, slots, thr, B, n_cand, min_data,
                       cosine, sc, path_len, path_slot, path_val, path_bin, scores, parent, cand_w, cand_ref, is_root, part_v, part_i, slot0,
                       keep_derived ? 1 : 0, part_s, part_n, cand_nr);
}

int argmax_parts(int n_cand) { return (n_cand + kArgmaxThreads - 1) / kArgmaxThreads; }
void argmax(const float *scores, int n_nodes, int n_cand, const float *w, const int32_t *ref, const float *parent,
            const int32_t *is_root, bool oblivious, float *part_v, int32_t *part_i, int32_t *best_idx, float *best_score,
            hipStream_t s, float *part_s) {
    const int parts = argmax_parts(n_cand);
    const int out_nodes = oblivious ? 1 : n_nodes;
    hipLaunchKernelGGL(k_argmax_stage1, dim3(parts, out_nodes), dim3(kArgmaxThreads), 0, s, scores, n_nodes, n_cand, w, ref, parent,
                       is_root, oblivious ? 1 : 0, part_v, part_i, part_s);
    (void)best_idx; (void)best_score;   // published by k_resolve_splits, which runs the last reduction stage itself
}
void resolve_splits(const float *part_v, const int32_t *part_i, int n_parts, int32_t *best_idx, float *best_score, bool oblivious, int n_nodes, const int32_t *ref_to_internal, const int32_t *cand_slot,
                    const FeatureSlot *slots, const int64_t *hist_local, const int64_t *hist_global, int Fp, int NB, int D,
                    NodeSplit *out, int64_t *counts4, int max_front, const int32_t *seg_start, int32_t *cursors, const uint32_t *thr_keys,
                    int B, hipStream_t s, void *pub, uint32_t *pub_flag, uint32_t pub_seq, unsigned *pub_done, const int64_t *hist_prev,
                    const int32_t *sub_par, const int32_t *sub_sib, const NearDetect *near) {
    hipLaunchKernelGGL(k_resolve_splits, dim3(n_nodes), dim3(64), 0, s, part_v, part_i, n_parts, best_idx, best_score, oblivious ? 1 : 0,
                       ref_to_internal, cand_slot, slots, hist_local, hist_global, Fp, NB, D, out, counts4, max_front, seg_start, cursors, thr_keys, B,
                       static_cast<char *>(pub), pub_flag, pub_seq, pub_done, hist_prev, sub_par, sub_sib,
                       near ? near->part_s : nullptr, near ? near->part_n : nullptr, near ? near->rel : 0.0f, near ? near->parent : nullptr, near ? near->is_root : nullptr, near ? near->cosine : 1, near ? near->rows : 0);
}

// Row-sharded runs: k_resolve_splits wrote GLOBAL left sizes; the partition needs this rank's.
__global__ void k_localize_splits(NodeSplit *__restrict__ splits, const int32_t *__restrict__ n_local,
                                  const int64_t *__restrict__ right_local, int n_nodes) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_nodes) splits[k].n_left = n_local[k] - static_cast<int32_t>(right_local[k]);
}
void localize_splits(NodeSplit *splits, const int32_t *n_local, const int64_t *right_local, int n_nodes, hipStream_t s) {
    hipLaunchKernelGGL(k_localize_splits, dim3((n_nodes + 63) / 64), dim3(64), 0, s, splits, n_local, right_local, n_nodes);
}
// the two in one launch (row-sharded levels: global left sizes -> this rank's, then the level's result block goes to the host)
__global__ __launch_bounds__(256) void k_localize_publish(NodeSplit *__restrict__ splits, const int32_t *__restrict__ n_local, const int64_t *__restrict__ right_local,
                                                          int n_nodes, const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int words, uint32_t *flag, uint32_t seq) {
    for (int k = threadIdx.x; k < n_nodes; k += 256) splits[k].n_left = n_local[k] - static_cast<int32_t>(right_local[k]);
    for (int i = threadIdx.x; i < words; i += 256) __builtin_nontemporal_store(src[i], &dst[i]);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void localize_publish(NodeSplit *splits, const int32_t *n_local, const int64_t *right_local, int n_nodes, void *d_src, void *h_dst_mapped, size_t bytes,
                      uint32_t *flag_mapped, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(k_localize_publish, dim3(1), dim3(256), 0, s, splits, n_local, right_local, n_nodes, static_cast<const uint32_t *>(d_src),
                       static_cast<uint32_t *>(h_dst_mapped), static_cast<int>(bytes / 4), flag_mapped, seq);
}
// Copies `n` contiguous node histograms into their level slots (dst slot = slot_map[k]).
__global__ void k_hist_place(const int64_t *__restrict__ src, int64_t *__restrict__ dst, const int32_t *__restrict__ slot_map,
                             size_t node_elems) {
    const int32_t slot = slot_map[blockIdx.y];
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < node_elems;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        dst[static_cast<size_t>(slot) * node_elems + i] = src[static_cast<size_t>(blockIdx.y) * node_elems + i];
}
void hist_place(const int64_t *src, int64_t *dst, const int32_t *slot_map, int n, size_t node_elems, hipStream_t s) {
    const int bx = static_cast<int>(std::min<size_t>(256, (node_elems + 255) / 256));
    hipLaunchKernelGGL(k_hist_place, dim3(bx, n), dim3(256), 0, s, src, dst, slot_map, node_elems);
}

void count_right(const int32_t *rows, const uint16_t *codes, const uint32_t *kt, int n_rows, const Chunk *chunks, int n_chunks,
                 const NodeSplit *splits, int64_t *n_right, hipStream_t s) {
    hipLaunchKernelGGL(k_count_right, dim3(n_chunks), dim3(kPartThreads), 0, s, rows, codes, kt, n_rows, chunks, splits, n_right);
}

// One block, compact code (no unrolling: a single wave pays every instruction-cache miss itself).  Node counts live in LDS (a dependent chain of global loads costs ~1 us per link: the first, single-threaded version of
// this kernel took 70 us per level); thread 0 does the short serial parts (chunk length search, prefix sums over <= 1024 nodes), all
// threads write the tables.
constexpr int kPlanThreads = 256;
constexpr int kPlanMaxNodes = 1024;
__global__ __launch_bounds__(kPlanThreads) void k_plan_oblivious(int L, int n_rows, int chunk_rows, int budget, const NodeSplit *__restrict__ prev,
                                                                 const float *__restrict__ best_prev, const float *__restrict__ thr, int B, ObliviousPlan pl) {
    __shared__ int s_seg[kPlanMaxNodes], s_cnt[kPlanMaxNodes], s_small[kPlanMaxNodes / 2 + 1], s_cb[kPlanMaxNodes / 2 + 2], s_pb[kPlanMaxNodes + 1];
    __shared__ int s_stopped, s_t, s_cslot[kMaxPath], s_cbin[kMaxPath];
    __shared__ float s_cval[kMaxPath];
    const int tid = threadIdx.x;
    const int n_act = 1 << L, n_comp = L == 0 ? 1 : n_act / 2;
    if (tid == 0) {
        int stopped = 0;
        if (L > 0) {
            stopped = pl.state[0] != 0;
            if (!stopped && best_prev[0] == -INFINITY) { stopped = 1; pl.state[0] = 1; pl.state[1] = L - 1; }   // no split at level L-1
        } else {
            pl.state[0] = 0; pl.state[1] = -1;
        }
        s_stopped = stopped;
    }
#pragma unroll 1
    for (int q = tid; q < L - 1; q += kPlanThreads) { s_cslot[q] = pl.cond_slot[q]; s_cbin[q] = pl.cond_bin[q]; s_cval[q] = pl.cond_val[q]; }
    __syncthreads();
    const bool stopped = s_stopped != 0;
    // nodes of this level from the previous level's resolved splits
    if (L == 0) {
        if (tid == 0) { s_seg[0] = 0; s_cnt[0] = n_rows; }
    } else {
        const int32_t *pseg = pl.node_seg + static_cast<size_t>(L - 1) * pl.mf, *pcnt = pl.node_n + static_cast<size_t>(L - 1) * pl.mf;
    #pragma unroll 1
        for (int k = tid; k < n_act / 2; k += kPlanThreads) {
            const int nl = stopped ? 0 : prev[k].n_left, ps = stopped ? 0 : pseg[k], pc = stopped ? 0 : pcnt[k];
            s_seg[2 * k] = ps; s_cnt[2 * k] = nl;
            s_seg[2 * k + 1] = ps + nl; s_cnt[2 * k + 1] = pc - nl;
        }
        if (tid == 0 && !stopped) {   // the condition chosen at level L-1 (one per level)
            const NodeSplit c = prev[0];
            s_cslot[L - 1] = c.fslot;
            s_cbin[L - 1] = c.bin;
            s_cval[L - 1] = c.is_cat ? INFINITY : thr[static_cast<size_t>(c.fslot) * B + c.bin];
            pl.cond_slot[L - 1] = s_cslot[L - 1]; pl.cond_bin[L - 1] = s_cbin[L - 1]; pl.cond_val[L - 1] = s_cval[L - 1];
        }
    }
    __syncthreads();
    int32_t *seg = pl.node_seg + static_cast<size_t>(L) * pl.mf, *cnt = pl.node_n + static_cast<size_t>(L) * pl.mf;
#pragma unroll 1
    for (int k = tid; k < n_act; k += kPlanThreads) { seg[k] = s_seg[k]; cnt[k] = s_cnt[k]; pl.seg_starts[k] = s_seg[k]; }
    // the accumulated child of every pair (fewer rows; ties -> the left child); the other one is parent - sibling
    if (L == 0) {
        if (tid == 0) { s_small[0] = 0; pl.slot_map[0] = 0; pl.sub_par[0] = -1; pl.sub_sib[0] = -1; }
    } else {
    #pragma unroll 1
        for (int p = tid; p < n_comp; p += kPlanThreads) {
            const int l = 2 * p, r = 2 * p + 1;
            const int small = s_cnt[l] <= s_cnt[r] ? l : r, big = small == l ? r : l;
            s_small[p] = small;
            pl.slot_map[p] = small;
            pl.sub_par[small] = -1; pl.sub_sib[small] = -1;
            pl.sub_par[big] = stopped ? -1 : p; pl.sub_sib[big] = stopped ? -1 : small;
        }
    }
    __syncthreads();
    // histogram chunks: the smallest chunk length t in [1024, chunk_rows] for which the accumulated nodes need at most `budget` chunks
    // (chunk_rows when even that is too many), every node in equal parts (the host's balanced_chunk_rows + make_chunks).  Integer
    // divisions are slow and a serial bisection costs tens of microseconds on one thread: the threads evaluate 256 candidate lengths
    // at once, twice (coarse grid, then every integer inside the winning interval) -- the same t as the bisection, parts(t) is monotone.
    auto parts_at = [&](int t) {
        int q = 0;
#pragma unroll 1
        for (int p = 0; p < n_comp; ++p) q += (s_cnt[s_small[p]] + t - 1) / t;
        return q;
    };
    {
        const int lo = min(1024, chunk_rows), hi = chunk_rows;
        const int step = max(1, (hi - lo + kPlanThreads - 1) / kPlanThreads);
        if (tid == 0) s_t = hi;
        __syncthreads();
        const int tc = min(hi, lo + tid * step);                 // coarse grid (the last threads repeat hi)
        if (parts_at(tc) <= budget) atomicMin(&s_t, tc);
        __syncthreads();
        const int coarse = s_t;
        __syncthreads();
        if (coarse > lo) {                                        // refine inside (coarse - step, coarse]
            const int tf = coarse - step + 1 + tid;
            if (tf >= lo && tf < coarse && parts_at(tf) <= budget) atomicMin(&s_t, tf);
        }
        __syncthreads();
    }
#pragma unroll 1
    for (int p = tid; p < n_comp; p += kPlanThreads) s_cb[p + 1] = (s_cnt[s_small[p]] + s_t - 1) / s_t;
#pragma unroll 1
    for (int k = tid; k < n_act; k += kPlanThreads) s_pb[k + 1] = (s_cnt[k] + kPartitionRows - 1) / kPartitionRows;
    __syncthreads();
    if (tid == 0) {   // prefixes (additions only)
        int nc = 0;
        s_cb[0] = 0;
    #pragma unroll 1
        for (int p = 0; p < n_comp; ++p) { nc = min(pl.cap_h, nc + s_cb[p + 1]); s_cb[p + 1] = nc; }
        int np = 0;
        s_pb[0] = 0;
    #pragma unroll 1
        for (int k = 0; k < n_act; ++k) { np = min(pl.cap_p, np + s_pb[k + 1]); s_pb[k + 1] = np; }
    }
    __syncthreads();
    const int t = s_t;
#pragma unroll 1
    for (int p = tid; p <= n_comp; p += kPlanThreads) pl.chunk_begin[p] = s_cb[p];
    // one thread per chunk ENTRY (a node of 2^20 rows has 256 partition chunks: written by one thread they cost 25 us): the entry's
    // node is found by bisection on the prefix, its offset follows from its rank inside the node; entries past the end get len 0
#pragma unroll 1
    for (int i = tid; i < pl.cap_h; i += kPlanThreads) {
        Chunk ck{0, 0, 0, 0};
        if (i < s_cb[n_comp]) {
            int lo = 0, hi = n_comp - 1;
#pragma unroll 1
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_cb[mid] <= i) lo = mid; else hi = mid - 1; }
            const int p = lo, n = s_cnt[s_small[p]], parts = (n + t - 1) / t, each = parts ? (n + parts - 1) / parts : 0;
            const int off = (i - s_cb[p]) * each;
            if (off < n) ck = Chunk{p, s_seg[s_small[p]] + off, min(each, n - off), 0};
        }
        pl.chunks[i] = ck;
    }
#pragma unroll 1
    for (int i = tid; i < pl.cap_p; i += kPlanThreads) {
        Chunk ck{0, 0, 0, 0};
        if (i < s_pb[n_act]) {
            int lo = 0, hi = n_act - 1;
#pragma unroll 1
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pb[mid] <= i) lo = mid; else hi = mid - 1; }
            const int k = lo, n = s_cnt[k], parts = (n + kPartitionRows - 1) / kPartitionRows, each = parts ? (n + parts - 1) / parts : 0;
            const int off = (i - s_pb[k]) * each;
            if (off < n) ck = Chunk{k, s_seg[k] + off, min(each, n - off), 0};
        }
        pl.part_chunks[i] = ck;
    }
    // paths: every node of an oblivious level has the same conditions behind it
#pragma unroll 1
    for (int k = tid; k < n_act; k += kPlanThreads) { pl.path_len[k] = stopped ? 0 : L; pl.is_root[k] = L == 0; }
#pragma unroll 1
    for (int i = tid; i < n_act * L; i += kPlanThreads) {
        const int k = i / L, q = i - k * L;
        pl.path_slot[k * kMaxPath + q] = s_cslot[q];
        pl.path_val[k * kMaxPath + q] = s_cval[q];
        pl.path_bin[k * kMaxPath + q] = s_cbin[q];
    }
}
void plan_oblivious_level(int level, int n_rows, int chunk_rows, int budget, const NodeSplit *resolved_prev, const float *best_score_prev,
                          const float *thr, int B, const ObliviousPlan &pl, hipStream_t s) {
    hipLaunchKernelGGL(k_plan_oblivious, dim3(1), dim3(kPlanThreads), 0, s, level, n_rows, chunk_rows, budget, resolved_prev, best_score_prev, thr, B, pl);
}

void partition_rows(const int32_t *rows_in, int32_t *rows_out, const uint16_t *codes, const uint32_t *kt, int n_rows, const Chunk *chunks,
                    int n_chunks, const NodeSplit *splits, int32_t *cursors, hipStream_t s) {
    hipLaunchKernelGGL(k_partition, dim3(n_chunks), dim3(kPartThreads), 0, s, rows_in, rows_out, codes, kt, n_rows, chunks, splits,
                       cursors);
}

void leaf_sums(const float *grads, int D, const int32_t *rows, const Chunk *chunks, int n_chunks, const StepScales *sc, int64_t *acc,
               hipStream_t s) {
    const int bs = D <= 256 ? 256 : ((D + 63) / 64) * 64;
    hipLaunchKernelGGL(k_leaf_sums, dim3(n_chunks), dim3(bs), (D + 1) * sizeof(unsigned long long), s, grads, D, rows, chunks,
                       sc, acc);
}

}  // namespace kern
}  // namespace gbrl
