// quantile.hip -- exact multi-rank selection (A3) and binning on gfx950, column-major formulation.
//
// Problem: for each of F features, the data values at B given ranks (the reference sorts every column,
// fitter.cpp:77-90, and reads split_candidate_generator.cpp:216-249's ranks).  A full sort is ~100x more work than
// needed.  Here:
//   1. k_transpose_keys   obs [N][F] f32  ->  KT [F][N] u32, order-preserving keys (coalesced both ways through LDS)
//   2. k_sample_splitters per feature: jittered-stratified sample of <= 16384 keys, bitonic sort in LDS, the
//                         sorted sample (minus one key) is the splitter set (<= 4095 per feature)
//   3. k_class_count      class(x) = 2*#{splitters < x} + [x equals the next splitter]; exact class counts per feature.
//                         Odd ("equality") classes hold copies of ONE value, so heavy duplicates never inflate a list.
//   4. k_targets          per target rank: its class and rank inside the class; distinct open classes get a list
//   5. k_extract          second sweep: keys of listed classes are appended to their lists (~6 % of the data)
//   6. k_select           one wave per target: exact order statistic inside its (small) list by 32-step bisection
// Everything is exact; the sample only balances the classes.  If a list would overflow its budget the engine falls back
// to the 32-pass bisection of kernels.hip (same results, slower).
#include "kernels.h"
#include "hooks.h"
#include "kernels_common.h"
#include "small_prep.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace gbrl {
namespace kern {

namespace {

constexpr int kMaxSplit = 4095;
constexpr int kClasses = 2 * (kMaxSplit + 1);  // 8192

// ---- 1. transpose + key ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_keys(const float *__restrict__ obs, int n, int F, uint32_t *__restrict__ kt) {
    __shared__ uint32_t tile[64][65];
    const int r0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, f = f0 + tx;
        tile[i][tx] = (r < n && f < F) ? float_to_key(obs[static_cast<size_t>(r) * F + f]) : 0xffffffffu;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int f = f0 + i, r = r0 + tx;
        if (f < F && r < n) kt[static_cast<size_t>(f) * n + r] = tile[tx][i];
    }
}

// The same for F % 4 == 0 and n % 4 == 0 (the benchmark shape): 16-byte accesses on both sides -- a thread reads four keys of one
// row and writes four consecutive rows of one feature (4 + 4 memory instructions per thread instead of 16 + 16).
__global__ __launch_bounds__(256) void k_transpose_keys_v4(const float *__restrict__ obs, int n, int F, uint32_t *__restrict__ kt) {
    __shared__ uint32_t tile[64][65];
    const int r0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
    const int q = threadIdx.x & 15, p = threadIdx.x >> 4;   // 16 x 16
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + i * 16 + p, f = f0 + 4 * q;
        v[i] = (r < n && f < F) ? *reinterpret_cast<const float4 *>(obs + static_cast<size_t>(r) * F + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t *t = &tile[i * 16 + p][4 * q];
        t[0] = float_to_key(v[i].x); t[1] = float_to_key(v[i].y); t[2] = float_to_key(v[i].z); t[3] = float_to_key(v[i].w);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = f0 + i * 16 + p, r = r0 + 4 * q;
        if (f < F && r < n) {     // n % 4 == 0: the four rows exist together
            const uint4 o = make_uint4(tile[4 * q][i * 16 + p], tile[4 * q + 1][i * 16 + p], tile[4 * q + 2][i * 16 + p], tile[4 * q + 3][i * 16 + p]);
            *reinterpret_cast<uint4 *>(kt + static_cast<size_t>(f) * n + r) = o;
        }
    }
}

// ---- 2. sample + sort + splitters ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sample_splitters(const uint32_t *__restrict__ kt, int n, int S, int n_split,
                                                           uint32_t *__restrict__ splitters, uint32_t *__restrict__ splitters_bfs,
                                                           uint32_t *__restrict__ sample_out /*nullable: write the raw sample, no sort*/) {
    extern __shared__ uint32_t s[];  // [S], S power of two
    const int f = blockIdx.x;
    const uint32_t *col = kt + static_cast<size_t>(f) * n;
    if (n <= S) {
        for (int i = threadIdx.x; i < S; i += blockDim.x) s[i] = i < n ? col[i] : 0xffffffffu;
    } else {
        const double stride = static_cast<double>(n) / S;
        for (int i = threadIdx.x; i < S; i += blockDim.x) {
            const long long lo = static_cast<long long>(i * stride), hi = static_cast<long long>((i + 1) * stride);
            uint32_t h = (static_cast<uint32_t>(i) * 2654435761u) ^ (static_cast<uint32_t>(f) * 40503u + 0x9e3779b9u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const long long span = hi > lo ? hi - lo : 1;
            long long idx = lo + static_cast<long long>(h % static_cast<uint32_t>(span));
            if (idx >= n) idx = n - 1;
            s[i] = col[idx];
        }
    }
    __syncthreads();
    if (sample_out) {   // sharded runs: the ranks exchange their raw samples and sort the union (k_union_splitters)
        for (int i = threadIdx.x; i < S; i += blockDim.x) sample_out[static_cast<size_t>(f) * S + i] = s[i];
        return;
    }
    for (int k = 2; k <= S; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < S; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // splitter j = sample at position (j+1)*S/(n_split+1) - 1
    const int step = S / (n_split + 1);
    for (int j = threadIdx.x; j < n_split; j += blockDim.x) splitters[static_cast<size_t>(f) * kMaxSplit + j] = s[(j + 1) * step - 1];
    // the same splitters in breadth-first (Eytzinger) order: BFS index i at level L, position p holds sorted index
    // (2p+1) * 2^(levels-1-L) - 1.  A binary search over the SORTED array probes power-of-two strides, i.e. one LDS bank
    // (up to 32-way conflicts); in BFS order the probes of one level are contiguous.
    int levels = 0;
    while ((1 << levels) < n_split + 1) ++levels;
    for (int i = threadIdx.x; i < n_split; i += blockDim.x) {
        const int L = 31 - __clz(i + 1), pp = i + 1 - (1 << L);
        const int q = ((2 * pp + 1) << (levels - 1 - L)) - 1;
        splitters_bfs[static_cast<size_t>(f) * kMaxSplit + i] = s[(q + 1) * step - 1];
    }
}

// ---- small batches (RL-sized): the whole column fits in LDS -> sort it, read the ranks ---------------------------------------
// One block per feature: n <= S keys padded with the maximal key to S (a power of two <= 16384), bitonic sort in LDS,
// thr_keys[f][k] = sorted[cum[k] - 1].  One launch instead of the eight of the radix multi-select, which are launch-bound at
// these sizes.  Exact by construction (padding sorts last and no rank points into it).
// (small_prep.h: sort_quantiles_body, shared with the fused preparation kernel of small_prep.hip)
__global__ __launch_bounds__(1024) void k_sort_quantiles(const uint32_t *__restrict__ kt, int n, int S, const int64_t *__restrict__ cum, int B,
                                                         uint32_t *__restrict__ thr_keys, float *__restrict__ thr_floats, int F,
                                                         uint16_t *__restrict__ codes) {
    extern __shared__ uint32_t s[];   // [S] keys, then [B] selected thresholds
    sort_quantiles_body<false>(kt, n, S, cum, B, thr_keys, thr_floats, F, static_cast<int>(blockIdx.x), codes, s);
}

// Sharded runs: sort the union of all ranks' samples ([F][SU] int64 after the exchange, SU a power of two <= 32768) and
// take every (SU / (n_split+1))-th key: every rank computes the identical splitter set.
__global__ __launch_bounds__(1024) void k_union_splitters(const int64_t *__restrict__ uni, int SU, int n_split,
                                                          uint32_t *__restrict__ splitters, uint32_t *__restrict__ splitters_bfs) {
    extern __shared__ uint32_t s[];
    const int f = blockIdx.x;
    for (int i = threadIdx.x; i < SU; i += blockDim.x) s[i] = static_cast<uint32_t>(uni[static_cast<size_t>(f) * SU + i]);
    __syncthreads();
    for (int k = 2; k <= SU; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < SU; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const int step = SU / (n_split + 1);
    for (int j = threadIdx.x; j < n_split; j += blockDim.x) splitters[static_cast<size_t>(f) * kMaxSplit + j] = s[(j + 1) * step - 1];
    int levels = 0;
    while ((1 << levels) < n_split + 1) ++levels;
    for (int i = threadIdx.x; i < n_split; i += blockDim.x) {
        const int L = 31 - __clz(i + 1), pp = i + 1 - (1 << L);
        const int q = ((2 * pp + 1) << (levels - 1 - L)) - 1;
        splitters_bfs[static_cast<size_t>(f) * kMaxSplit + i] = s[(q + 1) * step - 1];
    }
}
// place this rank's [F][S] u32 sample into its slice of the zeroed exchange buffer [F][world*S] int64 (sum == gather)
__global__ void k_place_sample(const uint32_t *__restrict__ samp, int F, int S, int rank, int SU, int64_t *__restrict__ uni) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<size_t>(F) * S) return;
    const size_t f = i / S, j = i % S;
    uni[f * SU + static_cast<size_t>(rank) * S + j] = samp[i];
}
// local class counts (summed over chunks) as int64 for the exchange; and back
__global__ void k_counts_to_i64(const uint32_t *__restrict__ partial, int n_chunks, size_t fc, int64_t *__restrict__ out) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= fc) return;
    int64_t v = 0;
    for (int q = 0; q < n_chunks; ++q) v += partial[static_cast<size_t>(q) * fc + i];
    out[i] = v;
}
// distributed bisection step on the extracted lists: counts[t] = #{local keys of target t's list < trial_t}
__global__ __launch_bounds__(256) void k_select_count(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ tgt_off,
                                                      const uint32_t *__restrict__ tgt_len, const uint32_t *__restrict__ prefix,
                                                      int bit, int n_targets, int64_t *__restrict__ counts) {
    const int t = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    if (t >= n_targets) return;
    const uint32_t off = tgt_off[t];
    uint32_t c = 0;
    if (off < 0xfffffffeu) {
        const uint32_t trial = prefix[t] | (1u << bit);
        const uint32_t *keys = lists + off;
        const uint32_t m = tgt_len[t];
        for (uint32_t i = lane; i < m; i += kWave) c += keys[i] < trial ? 1u : 0u;
        for (int o = kWave / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
    }
    if (lane == 0) counts[t] = c;
}
__global__ void k_select_update(uint32_t *__restrict__ prefix, const int64_t *__restrict__ counts, const uint32_t *__restrict__ tgt_off,
                                const uint32_t *__restrict__ tgt_rank, int bit, int n_targets, uint32_t *__restrict__ thr_keys) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_targets) return;
    if (tgt_off[t] >= 0xfffffffeu) return;   // direct answer already in thr_keys
    uint32_t p = prefix[t];
    if (counts[t] < static_cast<int64_t>(tgt_rank[t])) p |= (1u << bit);
    prefix[t] = p;
    if (bit == 0) thr_keys[t] = p;
}

// class of a key: 2 * #{splitters < key} + (key equals the first splitter >= key).  `e` holds the n_split = 2^levels - 1
// splitters in BFS order; the first splitter >= key is the node of the last left turn of the descent.
__device__ __forceinline__ int classify(const uint32_t *e, int levels, uint32_t key) {
    int i = 0;
    uint32_t last_ge = 0;
    bool went_left = false;
    for (int l = 0; l < levels; ++l) {
        const uint32_t v = e[i];
        const bool lt = v < key;
        if (!lt) { last_ge = v; went_left = true; }
        i = 2 * i + 1 + (lt ? 1 : 0);
    }
    const int j = i - ((1 << levels) - 1);
    return 2 * j + ((went_left && last_ge == key) ? 1 : 0);
}
// KQ independent descents in flight per thread (the LDS latency of one dependent chain is ~12 x 64 cycles)
constexpr int KQ = 8;
__device__ __forceinline__ void classifyN(const uint32_t *e, int levels, const uint32_t (&key)[KQ], int (&cls)[KQ]) {
    int i[KQ];
    uint32_t ge[KQ];
    bool wl[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) { i[q] = 0; ge[q] = 0; wl[q] = false; }
    for (int l = 0; l < levels; ++l) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const uint32_t v = e[i[q]];
            const bool lt = v < key[q];
            if (!lt) { ge[q] = v; wl[q] = true; }
            i[q] = 2 * i[q] + 1 + (lt ? 1 : 0);
        }
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) cls[q] = 2 * (i[q] - ((1 << levels) - 1)) + ((wl[q] && ge[q] == key[q]) ? 1 : 0);
}

// ---- 3. class counts ----------------------------------------------------------------------------------------------
constexpr int kCountThreads = 512;
__global__ __launch_bounds__(kCountThreads) void k_class_count(const uint32_t *__restrict__ kt, int n, int n_split,
                                                               const uint32_t *__restrict__ splitters, int chunk,
                                                               uint32_t *__restrict__ partial /*[n_chunks][F][8192]*/) {
    __shared__ uint32_t sp[kMaxSplit + 1];
    __shared__ uint32_t cnt[kClasses];
    const int f = blockIdx.y;
    for (int i = threadIdx.x; i < n_split; i += kCountThreads) sp[i] = splitters[static_cast<size_t>(f) * kMaxSplit + i];
    for (int i = threadIdx.x; i < kClasses; i += kCountThreads) cnt[i] = 0;
    __syncthreads();
    const uint32_t *col = kt + static_cast<size_t>(f) * n;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    int levels = 0;
    while ((1 << levels) < n_split + 1) ++levels;
    int i = lo + threadIdx.x;
    for (; i + (KQ - 1) * kCountThreads < hi; i += KQ * kCountThreads) {
        uint32_t k[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) k[q] = col[i + q * kCountThreads];
        int cls[KQ];
        classifyN(sp, levels, k, cls);
#pragma unroll
        for (int q = 0; q < KQ; ++q) atomicAdd(&cnt[cls[q]], 1u);
    }
    for (; i < hi; i += kCountThreads) atomicAdd(&cnt[classify(sp, levels, col[i])], 1u);
    __syncthreads();
    // per-chunk partials, plain coalesced stores (no global atomics); k_targets and k_extract sum them
    uint32_t *dst = partial + (static_cast<size_t>(blockIdx.x) * gridDim.y + f) * kClasses;
    for (int c = threadIdx.x; c < kClasses; c += kCountThreads) dst[c] = cnt[c];
}

// ---- 4. targets -----------------------------------------------------------------------------------------------------
// Per feature: inclusive prefix over the 8192 class counts (summed over the chunk partials); target k (rank cum_k, 1-based)
// lies in the first class whose inclusive prefix >= cum_k.  Equality classes answer directly; each distinct open target
// class gets a contiguous list [off, off+len) in the extraction buffer (class_off[f][class], 0xffffffff = not extracted).
__global__ __launch_bounds__(1024) void k_targets(const uint32_t *__restrict__ partial, int n_chunks, int F,
                                                  const int64_t *__restrict__ global_counts /*nullable [F][8192]: sharded runs*/,
                                                  const uint32_t *__restrict__ splitters, const int64_t *__restrict__ cum, int B,
                                                  uint32_t *__restrict__ class_off /*[F][8192], preset 0xffffffff*/,
                                                  uint32_t *__restrict__ tgt_off /*[F][B] list offset or 0xffffffff*/,
                                                  uint32_t *__restrict__ tgt_len, uint32_t *__restrict__ tgt_rank,
                                                  uint32_t *__restrict__ thr_keys /*[F][B], written for direct answers*/,
                                                  uint32_t *__restrict__ alloc /*[0] next element offset*/, uint32_t max_elems,
                                                  uint32_t *__restrict__ overflow) {
    __shared__ uint32_t pre[kClasses];   // inclusive prefix of the class counts (n < 2^31 rows)
    __shared__ int32_t tcls[1024];       // class of each target of the current tile of targets
    const int f = blockIdx.x;
    uint32_t *coff = class_off + static_cast<size_t>(f) * kClasses;
    __shared__ uint32_t loc_cnt[kClasses];   // this rank's class counts (list lengths)
    for (int c = threadIdx.x; c < kClasses; c += blockDim.x) {
        uint32_t v = 0;
        for (int q = 0; q < n_chunks; ++q) v += partial[(static_cast<size_t>(q) * F + f) * kClasses + c];
        loc_cnt[c] = v;
        pre[c] = global_counts ? static_cast<uint32_t>(global_counts[static_cast<size_t>(f) * kClasses + c]) : v;
    }
    __syncthreads();
    {   // block-wide inclusive scan of 8192 values: 8 per thread
        const int base = threadIdx.x * 8;
        uint32_t loc[8], run = 0;
        for (int q = 0; q < 8; ++q) { run += pre[base + q]; loc[q] = run; }
        __shared__ uint32_t tot[1024];
        tot[threadIdx.x] = run;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t v = threadIdx.x >= o ? tot[threadIdx.x - o] : 0;
            __syncthreads();
            tot[threadIdx.x] += v;
            __syncthreads();
        }
        const uint32_t off = threadIdx.x ? tot[threadIdx.x - 1] : 0;
        for (int q = 0; q < 8; ++q) pre[base + q] = loc[q] + off;
        __syncthreads();
    }
    for (int k0 = 0; k0 < B; k0 += 1024) {
        const int k = k0 + threadIdx.x;
        int c = -1;
        uint32_t want = 0, before = 0;
        if (k < B) {
            want = static_cast<uint32_t>(cum[k]);
            int lo = 0, hi = kClasses - 1;
            while (lo < hi) {  // first class with inclusive prefix >= want
                const int mid = (lo + hi) >> 1;
                if (pre[mid] < want) lo = mid + 1; else hi = mid;
            }
            c = lo;
            before = c ? pre[c - 1] : 0;
        }
        tcls[threadIdx.x] = c;
        __syncthreads();
        if (k < B) {
            const size_t t = static_cast<size_t>(f) * B + k;
            tgt_rank[t] = want - before;
            if (c & 1) {
                thr_keys[t] = splitters[static_cast<size_t>(f) * kMaxSplit + (c >> 1)];
                tgt_off[t] = 0xffffffffu;
                tgt_len[t] = 0;
            } else {
                // targets are sorted by rank, so equal classes are adjacent: the first target of a class allocates its list
                // (a class shared with the last target of the previous tile was allocated there: coff is already set)
                const bool first = (threadIdx.x == 0) ? (coff[c] == 0xffffffffu) : (tcls[threadIdx.x - 1] != c);
                const uint32_t len = loc_cnt[c];
                if (first) {
                    const uint32_t off = atomicAdd(&alloc[0], len);
                    if (off + len > max_elems || off + len < off) { atomicExch(overflow, 1u); coff[c] = 0xfffffffeu; }
                    else coff[c] = off;
                }
                tgt_len[t] = len;
            }
        }
        __syncthreads();
        if (k < B && !(c & 1)) tgt_off[static_cast<size_t>(f) * B + k] = coff[c];
        __syncthreads();
    }
}

// ---- 5. extract -----------------------------------------------------------------------------------------------------
// Second sweep over the column chunk (same search).  The chunk's write position inside each list is known exactly from the
// partial counts (list offset + counts of the earlier chunks), so ranks inside the block come from LDS cursors: no global
// atomics.  (Storing the classes in pass 3 to skip this search was tried: the extra 256 MiB write cost more than it saved.)
__global__ __launch_bounds__(kCountThreads) void k_extract(const uint32_t *__restrict__ kt, int n, int n_split,
                                                           const uint32_t *__restrict__ splitters, int chunk,
                                                           const uint32_t *__restrict__ class_off,
                                                           const uint32_t *__restrict__ partial, uint32_t *__restrict__ out) {
    __shared__ uint32_t sp[kMaxSplit + 1];
    __shared__ uint32_t cur[kClasses];
    const int f = blockIdx.y, F = gridDim.y;
    for (int i = threadIdx.x; i < n_split; i += kCountThreads) sp[i] = splitters[static_cast<size_t>(f) * kMaxSplit + i];
    {   // cursor of every class = list offset + counts of the earlier chunks; all loads of a round are issued together
        constexpr int PER = kClasses / kCountThreads;
        uint32_t off[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) off[u] = class_off[static_cast<size_t>(f) * kClasses + threadIdx.x + u * kCountThreads];
        for (unsigned q = 0; q < blockIdx.x; ++q) {
            uint32_t add[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) add[u] = partial[(static_cast<size_t>(q) * F + f) * kClasses + threadIdx.x + u * kCountThreads];
#pragma unroll
            for (int u = 0; u < PER; ++u) off[u] += add[u];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t base = class_off[static_cast<size_t>(f) * kClasses + threadIdx.x + u * kCountThreads];
            cur[threadIdx.x + u * kCountThreads] = base < 0xfffffffeu ? off[u] : 0xffffffffu;
        }
    }
    __syncthreads();
    const uint32_t *col = kt + static_cast<size_t>(f) * n;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    int levels = 0;
    while ((1 << levels) < n_split + 1) ++levels;
    int i = lo + threadIdx.x;
    for (; i + (KQ - 1) * kCountThreads < hi; i += KQ * kCountThreads) {
        uint32_t k[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) k[q] = col[i + q * kCountThreads];
        int cls[KQ];
        classifyN(sp, levels, k, cls);
#pragma unroll
        for (int q = 0; q < KQ; ++q)
            if (cur[cls[q]] != 0xffffffffu) out[atomicAdd(&cur[cls[q]], 1u)] = k[q];
    }
    for (; i < hi; i += kCountThreads) {
        const uint32_t key = col[i];
        const int cls = classify(sp, levels, key);
        if (cur[cls] != 0xffffffffu) out[atomicAdd(&cur[cls], 1u)] = key;
    }
}

// ---- 6. select ------------------------------------------------------------------------------------------------------
// One wave per target: v = smallest key with #{keys <= v} >= r, built bit by bit (prefix p; trial t = p|bit; set the bit
// iff #{keys < t} < r).
__global__ __launch_bounds__(256) void k_select(const uint32_t *__restrict__ lists, const uint32_t *__restrict__ tgt_off,
                                                const uint32_t *__restrict__ tgt_len, const uint32_t *__restrict__ tgt_rank,
                                                int n_targets, uint32_t *__restrict__ thr_keys) {
    const int t = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    if (t >= n_targets) return;
    const uint32_t off = tgt_off[t];
    if (off >= 0xfffffffeu) return;  // answered directly by an equality class (or overflowed: the engine falls back)
    const uint32_t *keys = lists + off;
    const uint32_t m = tgt_len[t], r = tgt_rank[t];
    uint32_t p = 0;
    constexpr int R = 16;            // lists up to 64*R keys are held in registers for the 32 bisection steps
    if (m <= kWave * R) {
        uint32_t k[R];
#pragma unroll
        for (int q = 0; q < R; ++q) k[q] = (lane + q * kWave) < m ? keys[lane + q * kWave] : 0xffffffffu;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = p | (1u << bit);
            uint32_t c = 0;
#pragma unroll
            for (int q = 0; q < R; ++q) c += k[q] < trial ? 1u : 0u;   // pads (max key) are never < trial... unless trial is max: excluded below
            for (int o = kWave / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
            if (c < r) p = trial;
        }
    } else {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = p | (1u << bit);
            uint32_t c = 0;
            for (uint32_t i = lane; i < m; i += kWave) c += keys[i] < trial ? 1u : 0u;
            for (int o = kWave / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
            if (c < r) p = trial;
        }
    }
    if (lane == 0) thr_keys[t] = p;
}

// ---- binning from the transposed keys ---------------------------------------------------------------------------------
// codes[g][row][fl] (u16, fl fastest, 16 features per group) = #{k : thr_key[f][k] < key(row, f)}.  One block = one group x
// 256 rows: reads 16 column segments of 1 KiB (coalesced), searches the 16 threshold rows staged in LDS, and writes the
// 8 KiB code tile contiguously.
constexpr int kGroup = 16;
constexpr int kBinTiles = 16;   // most row tiles per block (amortises staging the 16 x 511 threshold tree); fewer for small batches
__global__ __launch_bounds__(256) void k_bin_cols(const uint32_t *__restrict__ kt, int n, int F, const uint32_t *__restrict__ thr,
                                                  int B, int levels, int tiles_per_block, uint16_t *__restrict__ codes) {
    extern __shared__ uint32_t lds[];
    const int P = (1 << levels) - 1;                      // thresholds padded with the maximal key to a full tree
    uint32_t *t = lds;                                    // [16][P] in BFS order
    uint16_t *tile_buf = reinterpret_cast<uint16_t *>(lds + kGroup * P);  // [256][16]
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < kGroup * P; i += 256) {
        const int fl = i / P, e = i % P, f = g * kGroup + fl;
        const int L = 31 - __clz(e + 1), pp = e + 1 - (1 << L);
        const int q = ((2 * pp + 1) << (levels - 1 - L)) - 1;   // sorted index held by BFS node e
        t[i] = (f < F && q < B) ? thr[static_cast<size_t>(f) * B + q] : 0xffffffffu;
    }
    __syncthreads();
    // the threshold tree is staged once and reused for kBinTiles tiles of 256 rows
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int r0 = (blockIdx.x * tiles_per_block + tile) * 256;
        if (r0 >= n) break;
        const int r = r0 + threadIdx.x;
        const int rr = r < n ? r : n - 1;
        uint32_t key[kGroup];
#pragma unroll
        for (int q = 0; q < kGroup; ++q) {                    // all 16 column loads in flight before the first descent
            const int f = g * kGroup + q;
            key[q] = kt[static_cast<size_t>(f < F ? f : F - 1) * n + rr];
        }
#pragma unroll
        for (int fl0 = 0; fl0 < kGroup; fl0 += 8) {           // 8 independent descents in flight
            int idx[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) idx[q] = 0;
            for (int l = 0; l < levels; ++l) {
#pragma unroll
                for (int q = 0; q < 8; ++q) idx[q] = 2 * idx[q] + 1 + ((t[(fl0 + q) * P + idx[q]] < key[fl0 + q]) ? 1 : 0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int f = g * kGroup + fl0 + q;
                int code = idx[q] - P;                        // #{padded thresholds < key}; pads are never < key
                if (code > B) code = B;
                tile_buf[threadIdx.x * kGroup + fl0 + q] = static_cast<uint16_t>((f < F && r < n) ? code : 0);
            }
        }
        __syncthreads();
        const int rows = min(256, n - r0);
        {   // the tile's 8 KiB of codes leave in 16-byte pieces (tile_buf and the group's plane are 32-byte aligned per row)
            uint4 *dst = reinterpret_cast<uint4 *>(codes + (static_cast<size_t>(g) * n + r0) * kGroup);
            const uint4 *src = reinterpret_cast<const uint4 *>(tile_buf);
            for (int i = threadIdx.x; i < rows * 2; i += 256) dst[i] = src[i];
        }
        __syncthreads();
    }
}

#define GBRL_BIN_STEP_L8_F0(U, T, K) \
    asm volatile("v_lshl_add_u32 %8, %0, 2, %24\n ds_read_b32 %8, %8 offset:0\n" \
                 "v_lshl_add_u32 %9, %1, 2, %24\n ds_read_b32 %9, %9 offset:1024\n" \
                 "v_lshl_add_u32 %10, %2, 2, %24\n ds_read_b32 %10, %10 offset:2048\n" \
                 "v_lshl_add_u32 %11, %3, 2, %24\n ds_read_b32 %11, %11 offset:3072\n" \
                 "v_lshl_add_u32 %12, %4, 2, %24\n ds_read_b32 %12, %12 offset:4096\n" \
                 "v_lshl_add_u32 %13, %5, 2, %24\n ds_read_b32 %13, %13 offset:5120\n" \
                 "v_lshl_add_u32 %14, %6, 2, %24\n ds_read_b32 %14, %14 offset:6144\n" \
                 "v_lshl_add_u32 %15, %7, 2, %24\n ds_read_b32 %15, %15 offset:7168\n" \
                 "s_waitcnt lgkmcnt(0)\n" \
                 "v_cmp_lt_u32 vcc, %8, %16\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n" \
                 "v_cmp_lt_u32 vcc, %9, %17\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n" \
                 "v_cmp_lt_u32 vcc, %10, %18\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n" \
                 "v_cmp_lt_u32 vcc, %11, %19\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n" \
                 "v_cmp_lt_u32 vcc, %12, %20\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n" \
                 "v_cmp_lt_u32 vcc, %13, %21\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n" \
                 "v_cmp_lt_u32 vcc, %14, %22\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n" \
                 "v_cmp_lt_u32 vcc, %15, %23\n v_addc_co_u32 %7, vcc, %7, %7, vcc\n" \
                 : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]), "+v"(U[4]), "+v"(U[5]), "+v"(U[6]), "+v"(U[7]), \
                   "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(T[6]), "=&v"(T[7]) \
                 : "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]), "s"(lds_base) \
                 : "vcc", "memory")
#define GBRL_BIN_STEP_L8_F8(U, T, K) \
    asm volatile("v_lshl_add_u32 %8, %0, 2, %24\n ds_read_b32 %8, %8 offset:8192\n" \
                 "v_lshl_add_u32 %9, %1, 2, %24\n ds_read_b32 %9, %9 offset:9216\n" \
                 "v_lshl_add_u32 %10, %2, 2, %24\n ds_read_b32 %10, %10 offset:10240\n" \
                 "v_lshl_add_u32 %11, %3, 2, %24\n ds_read_b32 %11, %11 offset:11264\n" \
                 "v_lshl_add_u32 %12, %4, 2, %24\n ds_read_b32 %12, %12 offset:12288\n" \
                 "v_lshl_add_u32 %13, %5, 2, %24\n ds_read_b32 %13, %13 offset:13312\n" \
                 "v_lshl_add_u32 %14, %6, 2, %24\n ds_read_b32 %14, %14 offset:14336\n" \
                 "v_lshl_add_u32 %15, %7, 2, %24\n ds_read_b32 %15, %15 offset:15360\n" \
                 "s_waitcnt lgkmcnt(0)\n" \
                 "v_cmp_lt_u32 vcc, %8, %16\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n" \
                 "v_cmp_lt_u32 vcc, %9, %17\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n" \
                 "v_cmp_lt_u32 vcc, %10, %18\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n" \
                 "v_cmp_lt_u32 vcc, %11, %19\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n" \
                 "v_cmp_lt_u32 vcc, %12, %20\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n" \
                 "v_cmp_lt_u32 vcc, %13, %21\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n" \
                 "v_cmp_lt_u32 vcc, %14, %22\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n" \
                 "v_cmp_lt_u32 vcc, %15, %23\n v_addc_co_u32 %7, vcc, %7, %7, vcc\n" \
                 : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]), "+v"(U[4]), "+v"(U[5]), "+v"(U[6]), "+v"(U[7]), \
                   "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(T[6]), "=&v"(T[7]) \
                 : "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]), "s"(lds_base) \
                 : "vcc", "memory")
#define GBRL_BIN_STEP_L9_F0(U, T, K) \
    asm volatile("v_lshl_add_u32 %8, %0, 2, %24\n ds_read_b32 %8, %8 offset:0\n" \
                 "v_lshl_add_u32 %9, %1, 2, %24\n ds_read_b32 %9, %9 offset:2048\n" \
                 "v_lshl_add_u32 %10, %2, 2, %24\n ds_read_b32 %10, %10 offset:4096\n" \
                 "v_lshl_add_u32 %11, %3, 2, %24\n ds_read_b32 %11, %11 offset:6144\n" \
                 "v_lshl_add_u32 %12, %4, 2, %24\n ds_read_b32 %12, %12 offset:8192\n" \
                 "v_lshl_add_u32 %13, %5, 2, %24\n ds_read_b32 %13, %13 offset:10240\n" \
                 "v_lshl_add_u32 %14, %6, 2, %24\n ds_read_b32 %14, %14 offset:12288\n" \
                 "v_lshl_add_u32 %15, %7, 2, %24\n ds_read_b32 %15, %15 offset:14336\n" \
                 "s_waitcnt lgkmcnt(0)\n" \
                 "v_cmp_lt_u32 vcc, %8, %16\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n" \
                 "v_cmp_lt_u32 vcc, %9, %17\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n" \
                 "v_cmp_lt_u32 vcc, %10, %18\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n" \
                 "v_cmp_lt_u32 vcc, %11, %19\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n" \
                 "v_cmp_lt_u32 vcc, %12, %20\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n" \
                 "v_cmp_lt_u32 vcc, %13, %21\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n" \
                 "v_cmp_lt_u32 vcc, %14, %22\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n" \
                 "v_cmp_lt_u32 vcc, %15, %23\n v_addc_co_u32 %7, vcc, %7, %7, vcc\n" \
                 : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]), "+v"(U[4]), "+v"(U[5]), "+v"(U[6]), "+v"(U[7]), \
                   "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(T[6]), "=&v"(T[7]) \
                 : "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]), "s"(lds_base) \
                 : "vcc", "memory")
#define GBRL_BIN_STEP_L9_F8(U, T, K) \
    asm volatile("v_lshl_add_u32 %8, %0, 2, %24\n ds_read_b32 %8, %8 offset:16384\n" \
                 "v_lshl_add_u32 %9, %1, 2, %24\n ds_read_b32 %9, %9 offset:18432\n" \
                 "v_lshl_add_u32 %10, %2, 2, %24\n ds_read_b32 %10, %10 offset:20480\n" \
                 "v_lshl_add_u32 %11, %3, 2, %24\n ds_read_b32 %11, %11 offset:22528\n" \
                 "v_lshl_add_u32 %12, %4, 2, %24\n ds_read_b32 %12, %12 offset:24576\n" \
                 "v_lshl_add_u32 %13, %5, 2, %24\n ds_read_b32 %13, %13 offset:26624\n" \
                 "v_lshl_add_u32 %14, %6, 2, %24\n ds_read_b32 %14, %14 offset:28672\n" \
                 "v_lshl_add_u32 %15, %7, 2, %24\n ds_read_b32 %15, %15 offset:30720\n" \
                 "s_waitcnt lgkmcnt(0)\n" \
                 "v_cmp_lt_u32 vcc, %8, %16\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n" \
                 "v_cmp_lt_u32 vcc, %9, %17\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n" \
                 "v_cmp_lt_u32 vcc, %10, %18\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n" \
                 "v_cmp_lt_u32 vcc, %11, %19\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n" \
                 "v_cmp_lt_u32 vcc, %12, %20\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n" \
                 "v_cmp_lt_u32 vcc, %13, %21\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n" \
                 "v_cmp_lt_u32 vcc, %14, %22\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n" \
                 "v_cmp_lt_u32 vcc, %15, %23\n v_addc_co_u32 %7, vcc, %7, %7, vcc\n" \
                 : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]), "+v"(U[4]), "+v"(U[5]), "+v"(U[6]), "+v"(U[7]), \
                   "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(T[6]), "=&v"(T[7]) \
                 : "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(K[5]), "v"(K[6]), "v"(K[7]), "s"(lds_base) \
                 : "vcc", "memory")

// k_bin_cols_fast<LEVELS> (round 4): the same class codes with 3 VALU instructions per probe instead of 5.  PMC (scripts/step_pmc.sh): k_bin_cols
// kept the VALU busy 79 % of its 224 us -- 13 300 VALU instructions per wave, 52 per key -- so it is bound by its instruction count, not by
// HBM (its 0.8 GB would take 0.16 ms) or LDS (16 %).  Here the threshold tree of a feature sits in LDS in 1-based heap order with a stride
// of 2^LEVELS words (element 0 unused), so that the descent is u <- 2u + (t[u] < key): ONE v_addc_co_u32 with the compare's carry, the
// address is one shift, and the feature's offset is the ds_read's immediate.  Eight descents run side by side inside one asm block (their
// eight LDS reads in flight together); the class is u - 2^LEVELS; two codes leave in one 32-bit word.  LEVELS = 8 / 9 cover 129..511
// thresholds per feature (n_bins 130..512); other sizes take k_bin_cols.  Same codes bit for bit (the selection tests and every golden
// fixture run through it; `GBRL_HIP_BIN_PLAIN=1` = k_bin_cols).
template <int LEVELS>
__global__ __launch_bounds__(256) void k_bin_cols_fast(const uint32_t *__restrict__ kt, int n, int F, const uint32_t *__restrict__ thr,
                                                       int B, int tiles_per_block, uint16_t *__restrict__ codes) {
    constexpr int S = 1 << LEVELS;
    extern __shared__ uint32_t lds[];
    uint32_t *t = lds;                                                     // [16][S]: heap node u of feature fl at fl * S + u, u = 1 .. S - 1
    uint32_t *tile_buf = lds + kGroup * S;                                 // [256][8] words = [256][16] codes
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(t));   // LDS byte address of t (low half of the flat address)
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < kGroup * S; i += 256) {
        const int fl = i >> LEVELS, u = i & (S - 1), f = g * kGroup + fl;
        uint32_t v = 0xffffffffu;
        if (u > 0) {
            const int L = 31 - __clz(u), pp = u - (1 << L);
            const int q = ((2 * pp + 1) << (LEVELS - 1 - L)) - 1;          // sorted index held by heap node u
            if (f < F && q < B) v = thr[static_cast<size_t>(f) * B + q];
        }
        t[i] = v;
    }
    __syncthreads();
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int r0 = (blockIdx.x * tiles_per_block + tile) * 256;
        if (r0 >= n) break;
        const int r = r0 + threadIdx.x;
        const int rr = r < n ? r : n - 1;
        uint32_t key[kGroup];
#pragma unroll
        for (int q = 0; q < kGroup; ++q) {
            const int f = g * kGroup + q;
            key[q] = kt[static_cast<size_t>(f < F ? f : F - 1) * n + rr];
        }
        uint32_t word[kGroup / 2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t u[8], tv[8], kk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { u[q] = 1; kk[q] = key[half * 8 + q]; }
#pragma unroll
            for (int l = 0; l < LEVELS; ++l) {
                if constexpr (LEVELS == 8) { if (half == 0) { GBRL_BIN_STEP_L8_F0(u, tv, kk); } else { GBRL_BIN_STEP_L8_F8(u, tv, kk); } }
                else                       { if (half == 0) { GBRL_BIN_STEP_L9_F0(u, tv, kk); } else { GBRL_BIN_STEP_L9_F8(u, tv, kk); } }
            }
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const int f0 = g * kGroup + half * 8 + q;
                uint32_t c0 = min(u[q] - S, static_cast<uint32_t>(B)), c1 = min(u[q + 1] - S, static_cast<uint32_t>(B));
                if (!(f0 < F && r < n)) c0 = 0;
                if (!(f0 + 1 < F && r < n)) c1 = 0;
                word[half * 4 + q / 2] = c0 | (c1 << 16);
            }
        }
        uint4 *row = reinterpret_cast<uint4 *>(tile_buf + threadIdx.x * 8);
        row[0] = make_uint4(word[0], word[1], word[2], word[3]);
        row[1] = make_uint4(word[4], word[5], word[6], word[7]);
        __syncthreads();
        const int rows = min(256, n - r0);
        {
            uint4 *dst = reinterpret_cast<uint4 *>(codes + (static_cast<size_t>(g) * n + r0) * kGroup);
            const uint4 *src = reinterpret_cast<const uint4 *>(tile_buf);
            for (int i = threadIdx.x; i < rows * 2; i += 256) dst[i] = src[i];
        }
        __syncthreads();
    }
}
#undef GBRL_BIN_STEP_L8_F0
#undef GBRL_BIN_STEP_L8_F8
#undef GBRL_BIN_STEP_L9_F0
#undef GBRL_BIN_STEP_L9_F8

__global__ void k_scatter_cat_codes_grouped(const uint16_t *__restrict__ cat_codes, int n, int Fc, int F,
                                            uint16_t *__restrict__ codes) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<size_t>(n) * Fc) return;
    const size_t r = i / Fc;
    const int slot = F + static_cast<int>(i % Fc);
    codes[(static_cast<size_t>(slot / kGroup) * n + r) * kGroup + (slot % kGroup)] = cat_codes[i];
}

}  // namespace

void transpose_keys(const float *obs, int n, int F, uint32_t *kt, hipStream_t s) {
    dim3 grid((n + 63) / 64, (F + 63) / 64);
    const bool plain = [] { const char *e = hooks::raw(hooks::TRANSPOSE_PLAIN); return e && e[0] == '1'; }();   // measurement hook
    if ((F & 3) == 0 && (n & 3) == 0 && !plain && (reinterpret_cast<uintptr_t>(obs) & 15) == 0 && (reinterpret_cast<uintptr_t>(kt) & 15) == 0)
        hipLaunchKernelGGL(k_transpose_keys_v4, grid, dim3(256), 0, s, obs, n, F, kt);
    else
        hipLaunchKernelGGL(k_transpose_keys, grid, dim3(256), 0, s, obs, n, F, kt);
}

QuantilePlan quantile_plan(int n) {
    QuantilePlan p;
    int S = 64;
    while (S < n && S < 4096) S <<= 1;    // whole column when n <= S, else a 4096-key jittered-stratified sample
    p.sample = S;
    p.n_split = S - 1 < kMaxSplit ? S - 1 : kMaxSplit;   // (n_split + 1) divides S: both are powers of two
    p.n_chunks = n >= (1 << 18) ? 8 : (n >= (1 << 15) ? 2 : 1);
    p.chunk = (n + p.n_chunks - 1) / p.n_chunks;
    return p;
}

void sample_splitters(const uint32_t *kt, int n, int F, const QuantilePlan &p, uint32_t *splitters, uint32_t *splitters_bfs, hipStream_t s) {
    static PerDeviceOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sample_splitters), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 1024); }
    hipLaunchKernelGGL(k_sample_splitters, dim3(F), dim3(1024), p.sample * sizeof(uint32_t), s, kt, n, p.sample, p.n_split, splitters, splitters_bfs,
                       static_cast<uint32_t *>(nullptr));
}

void sample_only(const uint32_t *kt, int n, int F, int S, uint32_t *sample_out, hipStream_t s) {
    static PerDeviceOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sample_splitters), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 1024); }
    hipLaunchKernelGGL(k_sample_splitters, dim3(F), dim3(1024), S * sizeof(uint32_t), s, kt, n, S, S - 1, static_cast<uint32_t *>(nullptr),
                       static_cast<uint32_t *>(nullptr), sample_out);
}
void place_sample(const uint32_t *samp, int F, int S, int rank, int SU, int64_t *uni, hipStream_t s) {
    const size_t tot = static_cast<size_t>(F) * S;
    hipLaunchKernelGGL(k_place_sample, dim3((tot + 255) / 256), dim3(256), 0, s, samp, F, S, rank, SU, uni);
}
void union_splitters(const int64_t *uni, int F, int SU, int n_split, uint32_t *splitters, uint32_t *splitters_bfs, hipStream_t s) {
    static PerDeviceOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_union_splitters), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    hipLaunchKernelGGL(k_union_splitters, dim3(F), dim3(1024), SU * sizeof(uint32_t), s, uni, SU, n_split, splitters, splitters_bfs);
}
void counts_to_i64(const uint32_t *partial, int n_chunks, size_t fc, int64_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_counts_to_i64, dim3((fc + 255) / 256), dim3(256), 0, s, partial, n_chunks, fc, out);
}
void select_count(const uint32_t *lists, const uint32_t *tgt_off, const uint32_t *tgt_len, const uint32_t *prefix, int bit, int n_targets,
                  int64_t *counts, hipStream_t s) {
    hipLaunchKernelGGL(k_select_count, dim3((n_targets + 3) / 4), dim3(256), 0, s, lists, tgt_off, tgt_len, prefix, bit, n_targets, counts);
}
void select_update(uint32_t *prefix, const int64_t *counts, const uint32_t *tgt_off, const uint32_t *tgt_rank, int bit, int n_targets,
                   uint32_t *thr_keys, hipStream_t s) {
    hipLaunchKernelGGL(k_select_update, dim3((n_targets + 255) / 256), dim3(256), 0, s, prefix, counts, tgt_off, tgt_rank, bit, n_targets, thr_keys);
}

void class_count(const uint32_t *kt, int n, int F, const QuantilePlan &p, const uint32_t *splitters, uint32_t *partial, hipStream_t s) {
    hipLaunchKernelGGL(k_class_count, dim3(p.n_chunks, F), dim3(kCountThreads), 0, s, kt, n, p.n_split, splitters, p.chunk, partial);
}

void quantile_targets(const uint32_t *partial, const int64_t *global_counts, const uint32_t *splitters, const int64_t *cum, int F, int B,
                      const QuantilePlan &p, uint32_t *class_off, uint32_t *tgt_off, uint32_t *tgt_len, uint32_t *tgt_rank,
                      uint32_t *thr_keys, uint32_t *alloc, uint32_t max_elems, uint32_t *overflow, hipStream_t s) {
    hipLaunchKernelGGL(k_targets, dim3(F), dim3(1024), 0, s, partial, p.n_chunks, F, global_counts, splitters, cum, B, class_off, tgt_off,
                       tgt_len, tgt_rank, thr_keys, alloc, max_elems, overflow);
}

void quantile_extract(const uint32_t *kt, int n, int F, const QuantilePlan &p, const uint32_t *splitters, const uint32_t *class_off,
                      const uint32_t *partial, uint32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_extract, dim3(p.n_chunks, F), dim3(kCountThreads), 0, s, kt, n, p.n_split, splitters, p.chunk, class_off,
                       partial, out);
}

void quantile_select(const uint32_t *lists, const uint32_t *tgt_off, const uint32_t *tgt_len, const uint32_t *tgt_rank, int n_targets,
                     uint32_t *thr_keys, hipStream_t s) {
    hipLaunchKernelGGL(k_select, dim3((n_targets + 3) / 4), dim3(256), 0, s, lists, tgt_off, tgt_len, tgt_rank, n_targets, thr_keys);
}

int sort_quantiles_max_rows() { return 4096; }
bool sort_quantiles_fits(int n, int B) { int S = 256; while (S < n) S <<= 1; return n <= sort_quantiles_max_rows() && (static_cast<size_t>(S) + B) * sizeof(uint32_t) <= 64 * 1024; }   // beyond this the bitonic sort (O(n log^2 n) in one block per feature) loses to the radix passes
void sort_quantiles(const uint32_t *kt, int n, int F, const int64_t *cum, int B, uint32_t *thr_keys, float *thr_floats, hipStream_t s, uint16_t *codes) {
    int S = 256;   // four keys per thread, whole waves: 256 keys per wave
    while (S < n) S <<= 1;
    static PerDeviceOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sort_quantiles), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); }
    // ((S + B) * 4 bytes of LDS: the engine takes this path only while that stays within 64 KiB -- sort_quantiles_fits)
    const int blocks = codes ? ((F + kCodeGroup - 1) / kCodeGroup) * kCodeGroup : F;   // with codes: the padding features of the last group too
    hipLaunchKernelGGL(k_sort_quantiles, dim3(blocks), dim3(S / 4), (static_cast<size_t>(S) + B) * sizeof(uint32_t), s, kt, n, S, cum, B, thr_keys, thr_floats, F, codes);
}

void bin_cols(const uint32_t *kt, int n, int F, const uint32_t *thr_keys, int B, uint16_t *codes, hipStream_t s) {
    int levels = 1;
    while ((1 << levels) - 1 < B) ++levels;
    const size_t lds = static_cast<size_t>(kGroup) * ((1 << levels) - 1) * sizeof(uint32_t) + 256 * kGroup * sizeof(uint16_t);
    static PerDeviceOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_cols), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    // enough blocks to fill the chip first (small, RL-sized batches), then up to kBinTiles tiles per block to amortise the staging
    const int groups = (F + kGroup - 1) / kGroup;
    const long long tiles = (static_cast<long long>(n) + 255) / 256;
    int tpb = static_cast<int>(std::min<long long>(kBinTiles, std::max<long long>(1, tiles * groups / 1024)));
    dim3 grid(static_cast<unsigned>((tiles + tpb - 1) / tpb), groups);
    const bool plain = [] { const char *e = hooks::raw(hooks::BIN_PLAIN); return e && e[0] == '1'; }();   /* read per call: the tests flip it */
    if (!plain && (levels == 8 || levels == 9)) {
        const size_t lds_fast = static_cast<size_t>(kGroup) * (1u << levels) * sizeof(uint32_t) + 256 * kGroup * sizeof(uint16_t);
        static PerDeviceOnce attr_fast;
        if (attr_fast.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_cols_fast<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_cols_fast<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        }
        if (levels == 8) hipLaunchKernelGGL(k_bin_cols_fast<8>, grid, dim3(256), lds_fast, s, kt, n, F, thr_keys, B, tpb, codes);
        else hipLaunchKernelGGL(k_bin_cols_fast<9>, grid, dim3(256), lds_fast, s, kt, n, F, thr_keys, B, tpb, codes);
        return;
    }
    hipLaunchKernelGGL(k_bin_cols, grid, dim3(256), lds, s, kt, n, F, thr_keys, B, levels, tpb, codes);
}

void scatter_cat_codes_grouped(const uint16_t *cat_codes, int n, int Fc, int F, uint16_t *codes, hipStream_t s) {
    const size_t tot = static_cast<size_t>(n) * Fc;
    hipLaunchKernelGGL(k_scatter_cat_codes_grouped, dim3((tot + 255) / 256), dim3(256), 0, s, cat_codes, n, Fc, F, codes);
}

}  // namespace kern
}  // namespace gbrl
