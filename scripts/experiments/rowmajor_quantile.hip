// rowmajor_quantile.hip -- quantile thresholds (A3) and class codes straight from the caller's ROW-MAJOR observation matrix.
//
// The feature-major pipeline (quantile.hip, radix_select.hip) copies the matrix into transposed keys (2 streams of 4 N F bytes),
// counts radix digits over them three times and reads them once more to bin: 6.8 streams of the key matrix per step.
// Here every pass reads obs [N][F] in 32-feature slabs (one full 128-byte line per row and block: measured 4.9-5.6 TB/s,
// scripts/slab_stream_bench.hip) and nothing is transposed:
//
//   1. k_rm_sample_map   per feature: 4096 sampled keys -> a MONOTONE map key -> bucket in [0, 1024): cell = top 8 bits of the
//                        order-preserving key (sign + 7 exponent bits), every cell gets buckets in proportion to its share of
//                        the sample and the low bits interpolate linearly inside the cell.  Any monotone map keeps the method
//                        exact; the sample only balances the bucket populations.
//   2. k_rm_count        one pass: exact bucket populations (uint16-pair counters in LDS, [32 features][1024]).
//   3. k_rm_targets      per feature: the bucket of every target rank; "target buckets" (those holding a rank, <= 256 of 1024);
//                        rank of each target among the keys of target buckets; class table bucket -> #thresholds below it.
//   4. k_rm_extract      second pass: the keys of target buckets (typically 10-25 % of the data) are appended to a per-feature
//                        list (write-combined in LDS, one global atomic per block, feature and 256 rows).
//   5. radix_select      the existing exact MSD radix multi-select runs on those lists (per-feature lengths and ranks): the
//                        thresholds are the same data values the feature-major pipeline finds.  Heavy duplicates / constant
//                        columns simply survive into the lists and are resolved there -- no fallback path.
//   6. k_rm_bin          third pass: code = #{thresholds < key} = class table[bucket] (+ a compare against the bucket's own
//                        thresholds for keys of target buckets); written in the production layout [slot/16][row][16] u16.
//
// Streams of the matrix: 3 reads + lists + codes, against transpose (R+W) + 3.2 counting passes + binning read.
//
// STATUS (round 3): EXPERIMENT, not part of the library.  It was wired into Engine::step behind GBRL_HIP_ROWMAJOR, passed the GPU
// suite (identical trees on the stress columns of test_fast_quantile_path_equals_bisection_path and the full-size split-choice
// tests) and measured at 2^20 x 128 on MI355X: k_rm_sample_map 19 us, k_rm_count 109 us, k_rm_targets 19 us, k_rm_extract 241 us,
// radix_select on the lists (25 % of the keys) 226 us, k_rm_bin 226 us, and k_partition +9 us per level (codes instead of the
// feature-major keys): candidates + binning + partition penalty 0.92 ms against 0.79 ms for the feature-major pipeline once its
// transpose counts the first radix digit itself (k_transpose_count).  The three row-major passes are VALU-bound (30-35 instructions
// per key), not HBM-bound, and the extraction + second selection cost what they replace.  Kept for the record ("Measured
// dead ends"); radix_select's per-column lengths / ranks that step 5 needs were removed again with it.
#include "kernels.h"
#include "kernels_common.h"

#include <algorithm>
#include <cstdlib>

namespace gbrl {
namespace kern {

namespace {

constexpr int kRmCellBits = 8;
constexpr int kRmCells = 1 << kRmCellBits;         // cells of the key space: sign + 7 exponent bits
constexpr int kRmBuckets = 1024;
constexpr int kRmSlab = 32;                        // features per block: 128-byte row pieces
constexpr int kRmThreads = 1024;
constexpr int kRmWaves = kRmThreads / 64;
constexpr int kRmChunk = 16384;                    // rows per block (< 65536: uint16 counters)
constexpr int kRmMaxChunks = 256;
constexpr int kRmLutStride = kRmCells + 1;         // words per feature (+1: spreads the features over the LDS banks)
constexpr int kRmCntStride = kRmBuckets / 2 + 1;   // counter words per feature (uint16 pairs)
constexpr int kRmTabStride = kRmBuckets + 2;       // class-table halfwords per feature
constexpr int kRmSample = 4096;
constexpr int kRmDepth = 2;                        // row groups a wave keeps in flight

__device__ __forceinline__ uint32_t rm_bucket(uint32_t key, uint32_t e) {
    const uint32_t b = (e >> 16) + (__umul24((key >> 8) & 0xffffu, e & 0xffffu) >> 16);
    return b < static_cast<uint32_t>(kRmBuckets) ? b : static_cast<uint32_t>(kRmBuckets - 1);
}

__device__ __forceinline__ uint32_t rm_block_scan_incl(uint32_t v, uint32_t *scratch /*[16]*/) {   // over kRmThreads threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d); if (lane >= d) v += t; }
    if (lane == 63) scratch[w] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < w; ++i) base += scratch[i];
    __syncthreads();
    return v + base;
}

// ---- 1. sample -> monotone bucket map ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rm_sample_map(const float *__restrict__ obs, int n, int F, uint32_t *__restrict__ lut) {
    __shared__ uint32_t cc[kRmCells];
    __shared__ uint32_t wsum[4];
    const int f = blockIdx.x, t = threadIdx.x;
    static_assert(kRmCells == 256, "one cell per thread");
    cc[t] = 0;
    __syncthreads();
    const int stride = n / kRmSample;              // >= 1 (the host requires n >= kRmSample)
    for (int i = t; i < kRmSample; i += 256) {
        const uint32_t h = (static_cast<uint32_t>(i) * 2654435761u + static_cast<uint32_t>(f) * 40503u) >> 8;
        const size_t r = static_cast<size_t>(i) * stride + h % static_cast<uint32_t>(stride);
        atomicAdd(&cc[float_to_key(obs[r * F + f]) >> (32 - kRmCellBits)], 1u);
    }
    __syncthreads();
    // a sampled cell gets 1 + its share of the remaining buckets: sum <= cells + (buckets - cells) = buckets
    const uint32_t c = cc[t];
    const uint32_t scale = c ? 1u + (c * static_cast<uint32_t>(kRmBuckets - kRmCells)) / static_cast<uint32_t>(kRmSample) : 0u;
    uint32_t v = scale;
    const int lane = t & 63, w = t >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(v, d); if (lane >= d) v += x; }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    uint32_t base = v - scale;
    for (int i = 0; i < w; ++i) base += wsum[i];
    lut[static_cast<size_t>(f) * kRmCells + t] = (base << 16) | scale;
}

// ---- shared slab plumbing ------------------------------------------------------------------------------------------------
// lane -> (row of the group, quad of features); a wave step covers 8 rows x 32 features with one float4 per lane.
struct RmLane {
    int lr, q;
};

__device__ __forceinline__ void rm_stage_lut(uint32_t *lds_lut, const uint32_t *__restrict__ lut, int f0, int F) {
    for (int i = threadIdx.x; i < kRmSlab * kRmCells; i += kRmThreads) {
        const int fl = i / kRmCells, c = i % kRmCells;
        lds_lut[fl * kRmLutStride + c] = f0 + fl < F ? lut[static_cast<size_t>(f0 + fl) * kRmCells + c] : 0u;
    }
}

// ---- 2. bucket populations -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRmThreads) void k_rm_count(const float *__restrict__ obs, int n, int F, const uint32_t *__restrict__ lut,
                                                         uint16_t *__restrict__ partial16, int n_chunks) {
    extern __shared__ uint32_t rm_lds[];
    uint32_t *llut = rm_lds;                                  // [32][kRmLutStride]
    uint32_t *cnt = rm_lds + kRmSlab * kRmLutStride;          // [32][kRmCntStride] uint16 pairs
    const int chunk = blockIdx.x, f0 = blockIdx.y * kRmSlab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    rm_stage_lut(llut, lut, f0, F);
    for (int i = tid; i < kRmSlab * kRmCntStride; i += kRmThreads) cnt[i] = 0;
    const int r_lo = chunk * kRmChunk, r_hi = min(n, r_lo + kRmChunk);
    const int lr = lane >> 3, q = lane & 7;
    const bool fq_ok = f0 + 4 * q < F;                        // F % 4 == 0
    const int fq = fq_ok ? f0 + 4 * q : 0;
    constexpr int kStep = 8 * kRmWaves;
    // clamped, UNCONDITIONAL loads: a load inside a branch is followed by `s_waitcnt vmcnt(0)` and the prefetch is gone
    auto load = [&](int r) -> float4 { return *reinterpret_cast<const float4 *>(obs + static_cast<size_t>(min(r, n - 1)) * F + fq); };
    int g = r_lo + wave * 8;
    float4 buf[kRmDepth];
#pragma unroll
    for (int d = 0; d < kRmDepth; ++d) buf[d] = load(g + d * kStep + lr);
    __syncthreads();
    const uint32_t *lq = llut + (4 * q) * kRmLutStride;
    uint32_t *cq = cnt + (4 * q) * kRmCntStride;
    for (; g < r_hi; g += kStep) {
        const float4 c = buf[0];
#pragma unroll
        for (int d = 0; d + 1 < kRmDepth; ++d) buf[d] = buf[d + 1];
        buf[kRmDepth - 1] = load(g + kRmDepth * kStep + lr);
        if (g + lr < r_hi && fq_ok) {
            const float v[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t key = float_to_key(v[j]);
                const uint32_t b = rm_bucket(key, lq[j * kRmLutStride + (key >> (32 - kRmCellBits))]);
                atomicAdd(cq + j * kRmCntStride + (b >> 1), 1u << ((b & 1u) * 16));
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < kRmSlab * (kRmBuckets / 2); i += kRmThreads) {
        const int fl = i / (kRmBuckets / 2), w = i % (kRmBuckets / 2);
        if (f0 + fl < F)
            reinterpret_cast<uint32_t *>(partial16)[(static_cast<size_t>(f0 + fl) * n_chunks + chunk) * (kRmBuckets / 2) + w] = cnt[fl * kRmCntStride + w];
    }
}

// ---- 3. target buckets, ranks among their keys, class table ----------------------------------------------------------------
// One block per feature, thread t = bucket t.  cum[k] = 1-based global rank of target k (non-decreasing).
__global__ __launch_bounds__(kRmThreads) void k_rm_targets(const uint16_t *__restrict__ partial16, int n_chunks, const int64_t *__restrict__ cum,
                                                           int B, uint32_t *__restrict__ bitmap /*[F][32]*/, uint32_t *__restrict__ n_list /*[F]*/,
                                                           int64_t *__restrict__ cum_f /*[F][B]*/, uint16_t *__restrict__ classtab /*[F][1024]*/,
                                                           uint32_t *__restrict__ chunk_off /*[F][n_chunks]: where chunk c's keys start in list f*/) {
    static_assert(kRmBuckets == kRmThreads, "one bucket per thread");
    __shared__ uint32_t incl[kRmBuckets], sincl[kRmBuckets], tcnt[kRmBuckets];
    __shared__ uint32_t scratch[16];
    __shared__ int tb[256];
    const int f = blockIdx.x, t = threadIdx.x;
    uint32_t c = 0;
    {
        const uint16_t *p = partial16 + static_cast<size_t>(f) * n_chunks * kRmBuckets + t;
        int ch = 0;
        for (; ch + 8 <= n_chunks; ch += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[static_cast<size_t>(ch + u) * kRmBuckets];
#pragma unroll
            for (int u = 0; u < 8; ++u) c += v[u];
        }
        for (; ch < n_chunks; ++ch) c += p[static_cast<size_t>(ch) * kRmBuckets];
    }
    const uint32_t ic = rm_block_scan_incl(c, scratch);
    incl[t] = ic;
    tcnt[t] = 0;
    __syncthreads();
    if (t < B) {
        const uint32_t rank = static_cast<uint32_t>(cum[t]);
        int lo = 0, hi = kRmBuckets - 1;          // first bucket whose inclusive count reaches the rank
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (incl[mid] < rank) lo = mid + 1; else hi = mid; }
        tb[t] = lo;
        atomicAdd(&tcnt[lo], 1u);
    }
    __syncthreads();
    const uint32_t nt = tcnt[t];                  // thresholds inside bucket t
    const uint32_t sc = nt ? c : 0u;
    const uint32_t sic = rm_block_scan_incl(sc, scratch);
    sincl[t] = sic;
    const uint32_t tbase = rm_block_scan_incl(nt, scratch) - nt;   // thresholds in buckets below t
    classtab[static_cast<size_t>(f) * kRmBuckets + t] = static_cast<uint16_t>(tbase | (nt ? 0x8000u : 0u));
    const unsigned long long m = __ballot(nt != 0);
    if ((t & 63) == 0) {
        bitmap[static_cast<size_t>(f) * (kRmBuckets / 32) + (t >> 5)] = static_cast<uint32_t>(m);
        bitmap[static_cast<size_t>(f) * (kRmBuckets / 32) + (t >> 5) + 1] = static_cast<uint32_t>(m >> 32);
    }
    if (t == kRmBuckets - 1) n_list[f] = sic;
    __syncthreads();
    // every block of the extraction pass writes its keys of target buckets into its own run of the list: the runs' starts are
    // prefix sums of the per-(chunk, feature) counts the first pass already produced -- no global atomics, fixed layout
    __shared__ uint32_t csum[kRmMaxChunks];
    tcnt[t] = nt ? 1u : 0u;                       // (tcnt is free again: target flag per bucket)
    __syncthreads();
    {
        const int wave = t >> 6, lane = t & 63;
        for (int ch = wave; ch < n_chunks; ch += kRmWaves) {
            const uint16_t *p = partial16 + (static_cast<size_t>(f) * n_chunks + ch) * kRmBuckets;
            uint32_t a = 0;
#pragma unroll
            for (int u = 0; u < kRmBuckets / 64; ++u) { const int b = u * 64 + lane; a += tcnt[b] ? p[b] : 0u; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) csum[ch] = a;
        }
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int ch = 0; ch < n_chunks; ++ch) { chunk_off[static_cast<size_t>(f) * n_chunks + ch] = run; run += csum[ch]; }
    }
    if (t < B) {
        const int b = tb[t];
        // keys below bucket b: incl[b] - count(b); of those, in target buckets: sincl[b] - count(b) (b itself is a target bucket)
        const uint32_t below_all = b ? incl[b - 1] : 0u, below_tgt = b ? sincl[b - 1] : 0u;
        cum_f[static_cast<size_t>(f) * B + t] = static_cast<int64_t>(static_cast<uint32_t>(cum[t]) - (below_all - below_tgt));
    }
}

// ---- 4. keys of target buckets -> per-feature lists ----------------------------------------------------------------------
// No staging and no block barrier in the loop: for each of its 8 keys per step a lane knows (ballot over the 8 lanes that hold the
// same feature) its rank among the wave's survivors of that feature; one LDS atomic per feature and instruction hands out the
// block's positions, and the key goes straight to list f at chunk_off[f][chunk] + position.
__global__ __launch_bounds__(kRmThreads) void k_rm_extract(const float *__restrict__ obs, int n, int F, const uint32_t *__restrict__ lut,
                                                           const uint32_t *__restrict__ bitmap, uint32_t *__restrict__ lists, size_t cap,
                                                           const uint32_t *__restrict__ chunk_off, int n_chunks) {
    extern __shared__ uint32_t rm_lds[];
    uint32_t *llut = rm_lds;                                          // [32][kRmLutStride]
    uint32_t *bmp = llut + kRmSlab * kRmLutStride;                    // [32][33]
    uint32_t *scur = bmp + kRmSlab * 33;                              // [32] next free position of the block's run in list f
    const int chunk = blockIdx.x, f0 = blockIdx.y * kRmSlab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    rm_stage_lut(llut, lut, f0, F);
    for (int i = tid; i < kRmSlab * 32; i += kRmThreads) {
        const int fl = i >> 5, w = i & 31;
        bmp[fl * 33 + w] = f0 + fl < F ? bitmap[static_cast<size_t>(f0 + fl) * 32 + w] : 0u;
    }
    if (tid < kRmSlab) scur[tid] = f0 + tid < F ? chunk_off[static_cast<size_t>(f0 + tid) * n_chunks + chunk] : 0u;
    const int r_lo = chunk * kRmChunk, r_hi = min(n, r_lo + kRmChunk);
    const int lr = lane >> 3, q = lane & 7;
    const bool fq_ok = f0 + 4 * q < F;
    const int fq = fq_ok ? f0 + 4 * q : 0;
    constexpr int kStep = 8 * kRmWaves;
    auto load = [&](int r) -> float4 { return *reinterpret_cast<const float4 *>(obs + static_cast<size_t>(min(r, n - 1)) * F + fq); };
    int g = r_lo + wave * 8;
    float4 buf[kRmDepth];
#pragma unroll
    for (int d = 0; d < kRmDepth; ++d) buf[d] = load(g + d * kStep + lr);
    __syncthreads();
    const uint32_t *lq = llut + (4 * q) * kRmLutStride;
    const unsigned long long same_q = 0x0101010101010101ull << q;                 // the 8 lanes (rows) that hold the same features
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (; g < r_hi; g += kStep) {
        const float4 c = buf[0];
#pragma unroll
        for (int d = 0; d + 1 < kRmDepth; ++d) buf[d] = buf[d + 1];
        buf[kRmDepth - 1] = load(g + kRmDepth * kStep + lr);
        const float v[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int fl = 4 * q + j;
            const uint32_t key = float_to_key(v[j]);
            const uint32_t b = rm_bucket(key, lq[j * kRmLutStride + (key >> (32 - kRmCellBits))]);
            const bool hit = fq_ok && g + lr < r_hi && ((bmp[fl * 33 + (b >> 5)] >> (b & 31)) & 1u);
            const unsigned long long m = __ballot(hit) & same_q;
            uint32_t base = 0;
            if (lr == 0 && m) base = atomicAdd(&scur[fl], static_cast<uint32_t>(__popcll(m)));   // lane q speaks for feature 4 q + j
            base = __shfl(base, q);
            if (hit) lists[static_cast<size_t>(f0 + fl) * cap + base + __popcll(m & below)] = key;
        }
    }
}

// ---- 6. class codes ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRmThreads) void k_rm_bin(const float *__restrict__ obs, int n, int F, const uint32_t *__restrict__ lut,
                                                       const uint16_t *__restrict__ classtab, const uint32_t *__restrict__ thr /*[F][B] keys*/,
                                                       int B, uint16_t *__restrict__ codes) {
    extern __shared__ uint32_t rm_lds[];
    uint32_t *llut = rm_lds;                                                      // [32][kRmLutStride]
    uint16_t *tab = reinterpret_cast<uint16_t *>(llut + kRmSlab * kRmLutStride);   // [32][kRmTabStride]
    uint32_t *lthr = reinterpret_cast<uint32_t *>(tab + kRmSlab * kRmTabStride);   // [32][B + 1], the last entry a sentinel
    const int chunk = blockIdx.x, f0 = blockIdx.y * kRmSlab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ts = B + 1;
    rm_stage_lut(llut, lut, f0, F);
    for (int i = tid; i < kRmSlab * kRmBuckets; i += kRmThreads) {
        const int fl = i / kRmBuckets, b = i % kRmBuckets;
        tab[fl * kRmTabStride + b] = f0 + fl < F ? classtab[static_cast<size_t>(f0 + fl) * kRmBuckets + b] : 0;
    }
    for (int i = tid; i < kRmSlab * ts; i += kRmThreads) {
        const int fl = i / ts, k = i % ts;
        lthr[i] = (f0 + fl < F && k < B) ? thr[static_cast<size_t>(f0 + fl) * B + k] : 0xffffffffu;   // sentinel: never < key
    }
    const int r_lo = chunk * kRmChunk, r_hi = min(n, r_lo + kRmChunk);
    const int lr = lane >> 3, q = lane & 7;
    const bool fq_ok = f0 + 4 * q < F;
    const int fq = fq_ok ? f0 + 4 * q : 0;
    constexpr int kStep = 8 * kRmWaves;
    auto load = [&](int r) -> float4 { return *reinterpret_cast<const float4 *>(obs + static_cast<size_t>(min(r, n - 1)) * F + fq); };
    int g = r_lo + wave * 8;
    float4 buf[kRmDepth];
#pragma unroll
    for (int d = 0; d < kRmDepth; ++d) buf[d] = load(g + d * kStep + lr);
    __syncthreads();
    const uint32_t *lq = llut + (4 * q) * kRmLutStride;
    const size_t grp = static_cast<size_t>(fq >> 4);
    uint16_t *cdst = codes + grp * n * kCodeGroup + (fq & 15);
    for (; g < r_hi; g += kStep) {
        const float4 c = buf[0];
#pragma unroll
        for (int d = 0; d + 1 < kRmDepth; ++d) buf[d] = buf[d + 1];
        buf[kRmDepth - 1] = load(g + kRmDepth * kStep + lr);
        if (g + lr < r_hi && fq_ok) {
            const float v[4] = {c.x, c.y, c.z, c.w};
            uint32_t code[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int fl = 4 * q + j;
                const uint32_t key = float_to_key(v[j]);
                const uint32_t b = rm_bucket(key, lq[j * kRmLutStride + (key >> (32 - kRmCellBits))]);
                const uint32_t e = tab[fl * kRmTabStride + b];
                uint32_t cd = e & 0x7fffu;
                if (e & 0x8000u) {
                    // the bucket holds thresholds: those of earlier buckets are below the key, those of later buckets above it;
                    // the bucket's own (and only they) need the comparison -- the walk stops at the first one that is not below
                    const uint32_t *tp = lthr + fl * ts;
                    while (tp[cd] < key) ++cd;
                }
                code[j] = cd;
            }
            *reinterpret_cast<uint2 *>(cdst + static_cast<size_t>(g + lr) * kCodeGroup) = make_uint2(code[0] | (code[1] << 16), code[2] | (code[3] << 16));
        }
    }
}

constexpr size_t rm_align(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

}  // namespace

bool rowmajor_quantile_ok(const float *obs, int n, int F, int B) {
    const char *e = std::getenv("GBRL_HIP_ROWMAJOR");   // measurement / test hook, read per call: 0 = the feature-major pipeline
    if (e && e[0] == '0') return false;
    const int chunks = (n + kRmChunk - 1) / kRmChunk;
    return n >= (1 << 16) && (n & 3) == 0 && (F & 3) == 0 && F > 0 && chunks <= kRmMaxChunks && B >= 1 && B <= 256 && B <= radix_max_targets() &&
           (reinterpret_cast<uintptr_t>(obs) & 15) == 0;
}

size_t rowmajor_scratch_bytes(int n, int F, int B) {
    const size_t f = static_cast<size_t>(F);
    const size_t chunks = static_cast<size_t>((n + kRmChunk - 1) / kRmChunk);
    return rm_align(f * kRmCells * 4) + rm_align(f * chunks * kRmBuckets * 2) + rm_align(f * 32 * 4 + 4) + rm_align(f * 4) + rm_align(f * B * 8) +
           rm_align(f * kRmBuckets * 2) + rm_align(f * chunks * 4);
}

// Thresholds: thr_keys[f][k] = key of 1-based rank cum[k] in column f of obs.  `lists` must hold F * n uint32 (the buffer the
// feature-major pipeline uses for its transposed keys).  radix_* buffers as for radix_select.  Returns radix_select's code.
int rowmajor_quantile_select(const float *obs, int n, int F, const int64_t *cum, int B, void *scratch, uint32_t *lists, void *radix_state,
                             uint32_t *radix_partial, uint32_t *radix_lists, uint32_t *thr_keys, hipStream_t s) {
    char *p = static_cast<char *>(scratch);
    auto take = [&](size_t bytes) { char *q = p; p += rm_align(bytes); return q; };
    const size_t f = static_cast<size_t>(F);
    const int chunks = (n + kRmChunk - 1) / kRmChunk;
    uint32_t *lut = reinterpret_cast<uint32_t *>(take(f * kRmCells * 4));
    uint16_t *partial16 = reinterpret_cast<uint16_t *>(take(f * chunks * kRmBuckets * 2));
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(take(f * 32 * 4 + 4));
    uint32_t *n_list = reinterpret_cast<uint32_t *>(take(f * 4));
    int64_t *cum_f = reinterpret_cast<int64_t *>(take(f * B * 8));
    (void)take(f * kRmBuckets * 2);   // class table (rowmajor_bin)
    uint32_t *chunk_off = reinterpret_cast<uint32_t *>(take(f * chunks * 4));
    const size_t lds_count = sizeof(uint32_t) * (kRmSlab * kRmLutStride + kRmSlab * kRmCntStride);
    const size_t lds_extract = sizeof(uint32_t) * (kRmSlab * kRmLutStride + kRmSlab * 33 + kRmSlab);
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rm_count), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_count));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rm_extract), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_extract));
    }
    const dim3 grid(chunks, (F + kRmSlab - 1) / kRmSlab);
    uint16_t *classtab = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(cum_f) + rm_align(f * B * 8));
    hipLaunchKernelGGL(k_rm_sample_map, dim3(F), dim3(256), 0, s, obs, n, F, lut);
    hipLaunchKernelGGL(k_rm_count, grid, dim3(kRmThreads), lds_count, s, obs, n, F, lut, partial16, chunks);
    hipLaunchKernelGGL(k_rm_targets, dim3(F), dim3(kRmThreads), 0, s, partial16, chunks, cum, B, bitmap, n_list, cum_f, classtab, chunk_off);
    hipLaunchKernelGGL(k_rm_extract, grid, dim3(kRmThreads), lds_extract, s, obs, n, F, lut, bitmap, lists, static_cast<size_t>(n), chunk_off, chunks);
    return radix_select(lists, n, F, nullptr, B, radix_state, radix_partial, radix_lists, thr_keys, s, nullptr, n_list, cum_f);
}

// Class codes of the numeric features from the same scratch (map + class table) and the thresholds radix_select wrote.
void rowmajor_bin(const float *obs, int n, int F, const void *scratch, const uint32_t *thr_keys, int B, uint16_t *codes, hipStream_t s) {
    const char *p = static_cast<const char *>(scratch);
    const size_t f = static_cast<size_t>(F);
    const int chunks = (n + kRmChunk - 1) / kRmChunk;
    const uint32_t *lut = reinterpret_cast<const uint32_t *>(p);
    p += rm_align(f * kRmCells * 4) + rm_align(f * chunks * kRmBuckets * 2) + rm_align(f * 32 * 4 + 4) + rm_align(f * 4) + rm_align(f * B * 8);
    const uint16_t *classtab = reinterpret_cast<const uint16_t *>(p);
    const size_t lds = sizeof(uint32_t) * (kRmSlab * kRmLutStride) + sizeof(uint16_t) * (kRmSlab * kRmTabStride) + sizeof(uint32_t) * kRmSlab * (B + 1);
    static PerDeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_rm_bin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_rm_bin, dim3(chunks, (F + kRmSlab - 1) / kRmSlab), dim3(kRmThreads), lds, s, obs, n, F, lut, classtab, thr_keys, B, codes);
}

}  // namespace kern
}  // namespace gbrl
