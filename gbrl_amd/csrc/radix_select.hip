// radix_select.hip -- exact multi-rank selection (A3) by MSD radix counting on the transposed keys.
//
// For every feature the B target ranks are refined together, most significant digit first, in four counting passes over
// the column (digits of 12 + 7 + 7 + 6 bits of the order-preserving 32-bit key).  After each pass a small per-feature
// kernel locates, for every target, the digit bucket that contains its rank, subtracts the ranks below it, and gives
// every DISTINCT prefix that still contains a target a slot (<= B of them); the next pass counts the following digit only
// for keys whose prefix owns a slot.  No sampling, no sorting, no data movement, no overflow case: exact for any
// distribution (heavy duplicates simply stay in one slot until the last digit).
//
//   counting pass p (k_radix_count<p>) : one block = one feature x one chunk of the column; LDS holds the prefix->slot maps
//        of the earlier digits and uint32 counters [slot][digit] (<= 256 x 128 = 128 KB); the block's counters are stored
//        as a partial (no global atomics, nothing to zero), summed by the target pass.
//   target pass   p (k_radix_targets)  : one block per feature, thread k = target k.
//
// Slot lookup is O(1): slot1 = map1[top 12 bits]; every slot keeps a 128-bit set of the next digits that still lead to a
// target and the id of its first child; child slot = first child + popcount(set bits below the digit).
#include "kernels.h"
#include "hooks.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace gbrl {
namespace kern {

namespace {

constexpr int kBins1 = 4096;              // 12-bit first digit
constexpr int kNone = 0xffff;
constexpr int kRadixThreads = 1024;
constexpr int kMaxTargets = 256;
constexpr int kChunks1 = 8;               // pass 1: small LDS footprint, many blocks
constexpr int kCopies1 = 4;               // pass 1: lane-interleaved private copies of the 4096-bin histogram (power of two)
constexpr int kChunksN = 8;               // upper bound of chunks in passes 2..4 (2 at F >= 128: one 128 KB block per CU)
constexpr int kFiltWords = 1024, kFiltShift = 17;   // hashed prefix filter of passes 3 / 4: 32768 bits
constexpr int kSlotStride = 128;          // digits per slot in the partials of passes 2..4

__host__ __device__ constexpr int radix_bins(int pass) { return pass == 1 ? kBins1 : (pass == 4 ? 64 : 128); }
__host__ __device__ constexpr int radix_shift(int pass) { return pass == 1 ? 20 : (pass == 2 ? 13 : (pass == 3 ? 6 : 0)); }

// Per-feature selection state in global memory (written by k_radix_targets, read by k_radix_count).
struct RadixState {
    uint16_t *map1;        // [F][4096]     first digit -> slot1 | kNone
    uint16_t *child_off;   // [F][2][258]   level L (0: slot1->slot2, 1: slot2->slot3): first child slot of parent slot s
    uint64_t *child_bits;  // [F][2][256][2] 128-bit set of the digits of parent s that lead to a child; the child's slot is
                           //               child_off[s] + (number of set bits below the digit): children are in digit order
    uint32_t *n_slots;     // [F][4]        slots alive after pass 1, 2, 3
    uint16_t *tgt_slot;    // [F][B]
    uint32_t *tgt_rank;    // [F][B]        1-based rank inside the slot
    uint32_t *tgt_prefix;  // [F][B]
    uint32_t *list_cnt;    // [F][chunks]   keys of the chunk that matched a live 19-bit prefix in pass 3
    uint32_t *lists;       // [F][chunks][cap] those keys (valid when list_cnt <= cap): pass 4 reads them instead of the column
    uint32_t list_cap;
};

// ---- counting pass --------------------------------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(kRadixThreads) void k_radix_count(const uint32_t *__restrict__ kt, int n, int n_chunks, int B,
                                                               RadixState st, uint32_t *__restrict__ partial, bool plain_loads) {
    extern __shared__ uint32_t rl[];
    constexpr int NB = radix_bins(PASS);
    constexpr int SH = radix_shift(PASS);
    const int f = blockIdx.y;
    // pass 1 keeps kCopies1 private copies of its (small) histogram, chosen by the lane: keys of real data concentrate on a few
    // first digits, and same-address LDS atomics of one wave serialise
    const int n_slots = PASS == 1 ? kCopies1 : static_cast<int>(st.n_slots[f * 4 + (PASS - 2)]);
    const int n_cnt = n_slots * NB;
    uint32_t *cnt = rl;                                                        // [n_slots][NB]
    uint16_t *map1 = reinterpret_cast<uint16_t *>(rl + (PASS == 1 ? kCopies1 * kBins1 : kMaxTargets * NB));   // [4096]
    uint64_t *cbits = reinterpret_cast<uint64_t *>(map1 + kBins1);             // [2][256][2]
    uint16_t *coff = reinterpret_cast<uint16_t *>(cbits + 2 * 512);            // [2][258]
    uint32_t *filt = reinterpret_cast<uint32_t *>(coff + 2 * 260);             // [1024] 32768-bit hashed set of the live prefixes
    uint32_t *queue = filt + kFiltWords + (threadIdx.x >> 6) * 128;            // [waves][128] filter hits waiting for the exact chain
    constexpr int PS = PASS == 3 ? 13 : 6;                                     // bits below the prefix a pass-3/4 key must match
    __shared__ uint32_t list_cursor;
    if (threadIdx.x == 0) list_cursor = 0;
    uint32_t *my_list = st.lists + (static_cast<size_t>(f) * n_chunks + blockIdx.x) * st.list_cap;
    for (int i = threadIdx.x; i < n_cnt; i += kRadixThreads) cnt[i] = 0;
    if (PASS >= 3) {
        // Few keys still match a live prefix (a few % in pass 3, ~0.1 % in pass 4): one bit test rejects the rest before the
        // exact slot chain.  No false negatives; false positives (~B / 32768) fall out of the chain.
        for (int i = threadIdx.x; i < kFiltWords; i += kRadixThreads) filt[i] = 0;
        __syncthreads();
        if (static_cast<int>(threadIdx.x) < B) {
            const uint32_t h = ((st.tgt_prefix[static_cast<size_t>(f) * B + threadIdx.x] >> PS) * 0x9E3779B1u) >> kFiltShift;
            atomicOr(&filt[h >> 5], 1u << (h & 31));
        }
    }
    if (PASS >= 2) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(st.map1 + static_cast<size_t>(f) * kBins1);
        for (int i = threadIdx.x; i < kBins1 / 2; i += kRadixThreads) reinterpret_cast<uint32_t *>(map1)[i] = src[i];
    }
    if (PASS >= 3) {
        const int levels = PASS - 2;
        for (int i = threadIdx.x; i < levels * 258; i += kRadixThreads) coff[i] = st.child_off[static_cast<size_t>(f) * 2 * 258 + i];
        for (int i = threadIdx.x; i < levels * 512; i += kRadixThreads) cbits[i] = st.child_bits[static_cast<size_t>(f) * 2 * 512 + i];
    }
    __syncthreads();
    const uint32_t *col = kt + static_cast<size_t>(f) * n;
    const int chunk = ((n + n_chunks - 1) / n_chunks + 3) & ~3;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    auto child = [&](int level, int slot, int digit) -> int {
        const uint64_t w0 = cbits[level * 512 + slot * 2], w1 = cbits[level * 512 + slot * 2 + 1];
        const uint64_t w = digit & 64 ? w1 : w0;
        const int bit = digit & 63;
        if (!((w >> bit) & 1)) return kNone;
        const int below = __popcll(w & ((1ull << bit) - 1)) + (digit & 64 ? __popcll(w0) : 0);
        return coff[level * 258 + slot] + below;
    };
    auto count_one = [&](uint32_t key) {
        int slot = PASS == 1 ? static_cast<int>(__lane_id() & (kCopies1 - 1)) : 0;
        if (PASS >= 2) {
            slot = map1[key >> 20];
            if (PASS >= 3 && slot != kNone) slot = child(0, slot, (key >> 13) & 127);
            if (PASS >= 4 && slot != kNone) slot = child(1, slot, (key >> 6) & 127);
        }
        const int idx = slot == kNone ? -1 : slot * NB + static_cast<int>((key >> SH) & (NB - 1));
        if (PASS == 3) {
            // the survivors of this pass (typically a few % of the column) are compacted into the block's own list, so that
            // the last pass does not stream the column again.  Wave-aggregated reservation on an LDS cursor.
            const unsigned long long m = __ballot(idx >= 0);
            if (m) {
                const int leader = __ffsll(static_cast<long long>(m)) - 1;
                uint32_t base = 0;
                if (static_cast<int>(__lane_id()) == leader) base = atomicAdd(&list_cursor, static_cast<uint32_t>(__popcll(m)));
                base = __builtin_amdgcn_readlane(base, leader);
                if (idx >= 0) {
                    const uint32_t pos = base + __popcll(m & ((1ull << __lane_id()) - 1));
                    if (pos < st.list_cap) my_list[pos] = key;
                }
            }
        }
        // a wave whose keys all fall into one counter (constant / mostly-constant columns) adds once instead of 64 times
        const int first = __builtin_amdgcn_readfirstlane(idx);
        const unsigned long long active = __ballot(1);
        const unsigned long long same = __ballot(idx == first);
        if (same == active) {
            if (first >= 0 && __lane_id() == static_cast<unsigned>(__ffsll(static_cast<long long>(active)) - 1))
                atomicAdd(&cnt[first], static_cast<uint32_t>(__popcll(active)));
        } else if (idx >= 0) {
            atomicAdd(&cnt[idx], 1u);
        }
    };
    constexpr int U = 16;   // keys per thread in flight (8: 0.441 ms, 16: 0.406 ms, 32: 0.432 ms for the four passes at 2^20 x 128)
    int i0 = lo + threadIdx.x;
    int qlen = 0;   // wave-uniform
    if (PASS == 4) {
        const uint32_t n_list = st.list_cnt[f * n_chunks + blockIdx.x];
        if (n_list <= st.list_cap) {   // the chunk's survivors of pass 3 fit their list: count those, skip the column
            for (uint32_t i = threadIdx.x; i < n_list; i += kRadixThreads) count_one(my_list[i]);
            i0 = hi;
        }
    }
    auto strip = [&](const uint32_t (&key)[U]) __attribute__((always_inline)) {
        if (PASS >= 3) {
            // The hits (a few % of the keys) are compacted ACROSS the wave into its LDS queue and the exact chain runs on 64 queued
            // keys at a time: per-lane handling made the whole wave walk the chain for one or two sparse candidates per strip.
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t h = ((key[u] >> PS) * 0x9E3779B1u) >> kFiltShift;
                const bool hit = (filt[h >> 5] >> (h & 31)) & 1;
                const unsigned long long m = __ballot(hit);
                if (m) {
                    if (hit) queue[qlen + __popcll(m & ((1ull << __lane_id()) - 1))] = key[u];
                    qlen += __popcll(m);
                    __builtin_amdgcn_wave_barrier();   // the queue is exchanged between the lanes of ONE wave: LDS order is program order
                    if (qlen >= 64) {
                        const uint32_t k0 = queue[__lane_id()];
                        const uint32_t k1 = queue[64 + __lane_id()];
                        count_one(k0);
                        qlen -= 64;
                        if (static_cast<int>(__lane_id()) < qlen) queue[__lane_id()] = k1;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) count_one(key[u]);
        }
    };
    if (i0 < hi && (n & 3) == 0 && !plain_loads) {
        // 16-byte loads (the column starts 16-byte aligned when n % 4 == 0, the chunk bounds are multiples of 4): a thread takes four
        // consecutive keys per load, two loads per iteration.  Counting does not depend on the order of the keys.
        const int iters = (hi - i0 + static_cast<int>(threadIdx.x)) / (kRadixThreads * U);      // (hi - chunk start) / keys per iteration
        const uint4 *c4 = reinterpret_cast<const uint4 *>(col + (i0 - static_cast<int>(threadIdx.x)));
        for (int it = 0; it < iters; ++it) {
            uint32_t key[U];
#pragma unroll
            for (int v = 0; v < U / 4; ++v) {
                const uint4 x = c4[(it * (U / 4) + v) * kRadixThreads + threadIdx.x];
                key[4 * v] = x.x; key[4 * v + 1] = x.y; key[4 * v + 2] = x.z; key[4 * v + 3] = x.w;
            }
            strip(key);
        }
        i0 += iters * kRadixThreads * U;
    }
    // full strips: straight-line loads.  The condition is evaluated for the wave's LAST lane, so that a wave enters a strip with all its
    // lanes or not at all: strip() compacts filter hits across the wave and keeps the queue length wave-uniform -- with a per-lane
    // condition the lanes that sat out kept a stale length and the wave's last queued keys were never counted (round 3: found by
    // scripts/selfcheck_sweep.py at N = 2^18 + 4, whose last chunk leaves 16 360 keys = a strip for lanes 0..999 only; passes 3 / 4)
    for (; i0 - static_cast<int>(__lane_id()) + 63 + (U - 1) * kRadixThreads < hi; i0 += kRadixThreads * U) {
        uint32_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u) key[u] = col[i0 + u * kRadixThreads];
        strip(key);
    }
    if (PASS >= 3 && static_cast<int>(__lane_id()) < qlen) count_one(queue[__lane_id()]);   // what is left in the wave's queue
    for (; i0 < hi; i0 += kRadixThreads) count_one(col[i0]);
    __syncthreads();
    if (PASS == 3 && threadIdx.x == 0) st.list_cnt[f * n_chunks + blockIdx.x] = list_cursor;
    uint32_t *dst = partial + (static_cast<size_t>(f) * n_chunks + blockIdx.x) * (PASS == 1 ? kBins1 : kMaxTargets * kSlotStride);
    if (PASS == 1) {
        for (int i = threadIdx.x; i < kBins1; i += kRadixThreads) {
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < kCopies1; ++c) v += cnt[c * kBins1 + i];
            dst[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < n_cnt; i += kRadixThreads) dst[(i / NB) * kSlotStride + (i % NB)] = cnt[i];
    }
}

// ---- transpose fused with the first counting pass ------------------------------------------------------------------------
// The first digit (top 12 bits of the key) of every key is counted WHILE the observation matrix is transposed into the
// feature-major keys: the keys pass through LDS anyway, and the separate first counting pass streamed the 4 N F bytes of keys once
// more (0.12 ms of the 0.41 ms selection at 2^20 x 128).  One block = 16 features x one chunk of <= kTcChunk rows; counters
// [16][4096] as uint16 pairs packed in uint32 words (128 KiB: a block sees < 65536 rows, so a half never carries into its
// neighbour) -- stored as they are, i.e. as a uint16 partial [feature][chunk][4096] that the pass-1 target kernel sums over chunks.
// The transposition is WAVE-private (no block barrier in the loop): a wave takes 16 rows x 16 features (one float4 per lane = 64-byte
// row pieces), turns them through its own 16 x 20-word LDS patch and stores, per feature, 16 consecutive rows = 64 bytes; the 16
// waves of a block work on adjacent row groups, so a block writes 1 KiB runs per feature.  Loads are clamped and UNCONDITIONAL: a
// load inside a branch is followed by `s_waitcnt vmcnt(0)` and the wave's prefetch is gone (scripts/slab_stream_bench.hip).
constexpr int kTcThreads = 1024;
constexpr int kTcWaves = kTcThreads / 64;
constexpr int kTcWaveRows = 16;              // rows per wave and step
constexpr int kTcPatch = 20;                 // words per feature row of a wave's patch: 16-byte aligned
constexpr int kTcChunk = 32768;              // rows per block (< 65536): 32 chunks x 8 slabs = 256 blocks at 2^20 x 128
constexpr int kTcMaxChunks = 128;            // partial buffer: F x chunks x 8 KiB <= radix_partial_bytes(F)
constexpr int kTcDepth = 2;                  // row groups a wave keeps in flight

__device__ __forceinline__ uint32_t tc_float_to_key(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u << 1) == 0) u = 0;
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kTcThreads) void k_transpose_count(const float *__restrict__ obs, int n, int F, uint32_t *__restrict__ kt,
                                                                uint16_t *__restrict__ partial16, int n_chunks) {
    extern __shared__ uint32_t tc_lds[];
    constexpr int CW = kBins1 / 2;                                            // counter words per feature
    uint32_t *cnt = tc_lds;                                                   // [16][CW] packed uint16 pairs
    const int chunk = blockIdx.x, f0 = blockIdx.y * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t *patch = tc_lds + 16 * CW + wave * (16 * kTcPatch);               // this wave's [16 features][kTcPatch]
    {
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        for (int i = tid; i < 16 * CW / 4; i += kTcThreads) c4[i] = make_uint4(0, 0, 0, 0);
    }
    const int r_lo = chunk * kTcChunk, r_hi = min(n, r_lo + kTcChunk);
    const int lr = lane >> 2, q = lane & 3;              // load: row of the group, quad of features
    const int sf = lane >> 2, sq = lane & 3;             // store: feature, four consecutive rows
    const bool fq_ok = f0 + 4 * q < F;                   // F % 4 == 0: a quad is inside or outside
    const int fq = fq_ok ? f0 + 4 * q : 0;
    constexpr int kStep = kTcWaveRows * kTcWaves;        // rows the block advances per step
    auto load = [&](int r) -> float4 { return *reinterpret_cast<const float4 *>(obs + static_cast<size_t>(min(r, n - 1)) * F + fq); };
    int g = r_lo + wave * kTcWaveRows;                   // first row of this wave's group
    float4 buf[kTcDepth];
#pragma unroll
    for (int d = 0; d < kTcDepth; ++d) buf[d] = load(g + d * kStep + lr);
    __syncthreads();                                     // counters zeroed
    uint32_t *cbase = cnt + (4 * q) * CW;
    for (; g < r_hi; g += kStep) {
        const float4 cur = buf[0];
#pragma unroll
        for (int d = 0; d + 1 < kTcDepth; ++d) buf[d] = buf[d + 1];
        buf[kTcDepth - 1] = load(g + kTcDepth * kStep + lr);
        const uint32_t k0 = tc_float_to_key(cur.x), k1 = tc_float_to_key(cur.y), k2 = tc_float_to_key(cur.z), k3 = tc_float_to_key(cur.w);
        uint32_t *t = patch + (4 * q) * kTcPatch + lr;
        t[0] = k0; t[kTcPatch] = k1; t[2 * kTcPatch] = k2; t[3 * kTcPatch] = k3;
        if (g + lr < r_hi && fq_ok) {
            // digit = key >> 20; word = digit / 2, half = digit & 1
            atomicAdd(cbase + (k0 >> 21), 1u << ((k0 >> 16) & 16));
            atomicAdd(cbase + CW + (k1 >> 21), 1u << ((k1 >> 16) & 16));
            atomicAdd(cbase + 2 * CW + (k2 >> 21), 1u << ((k2 >> 16) & 16));
            atomicAdd(cbase + 3 * CW + (k3 >> 21), 1u << ((k3 >> 16) & 16));
        }
        __builtin_amdgcn_wave_barrier();                 // the patch is exchanged between the lanes of ONE wave: LDS order is program order
        const uint4 o = *reinterpret_cast<const uint4 *>(patch + sf * kTcPatch + 4 * sq);
        __builtin_amdgcn_wave_barrier();
        if (f0 + sf < F && g + 4 * sq < r_hi)            // n % 4 == 0: four rows exist together
            *reinterpret_cast<uint4 *>(kt + static_cast<size_t>(f0 + sf) * n + g + 4 * sq) = o;
    }
    __syncthreads();
    // the block's counters leave as uint16 [feature][chunk][kBins1]
    for (int i = tid; i < 16 * CW / 4; i += kTcThreads) {
        const int fl = i / (CW / 4);
        if (f0 + fl < F) {
            uint4 *dst = reinterpret_cast<uint4 *>(partial16 + (static_cast<size_t>(f0 + fl) * n_chunks + chunk) * kBins1);
            dst[i % (CW / 4)] = reinterpret_cast<const uint4 *>(cnt)[i];
        }
    }
}

// ---- target pass -------------------------------------------------------------------------------------------------------
constexpr int kTgtThreads = 1024;

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d); if (lane >= d) v += t; }
    return v;
}
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t *scratch /*[16]*/) {   // over kTgtThreads threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_scan_incl(v);
    if (lane == 63) scratch[w] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < w; ++i) base += scratch[i];
    __syncthreads();
    return v + base;
}

// One block per feature; thread k < B owns target k.  `pass` = the counting pass being consumed.
__global__ __launch_bounds__(kTgtThreads) void k_radix_targets(int pass, const uint32_t *__restrict__ partial, int n_chunks,
                                                               const int64_t *__restrict__ cum, int B, RadixState st,
                                                               uint32_t *__restrict__ thr_keys, int p1_u16, uint32_t *__restrict__ le_out,
                                                               const int64_t *__restrict__ gbuf /* row-sharded: the exchanged counts, read in place */, int F) {
    extern __shared__ uint32_t sums[];        // inclusive digit counts: pass 1 [4096]; later [n_slots][NB]
    __shared__ int tslot[kMaxTargets], tdig[kMaxTargets];
    __shared__ uint32_t scratch[16], total_slots;
    const int f = blockIdx.x, k = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NB = radix_bins(pass);
    const int n_slots = pass == 1 ? 1 : static_cast<int>(st.n_slots[f * 4 + (pass - 2)]);
    const size_t pstride = pass == 1 ? kBins1 : kMaxTargets * kSlotStride;
    const uint32_t *p0 = partial + static_cast<size_t>(f) * n_chunks * pstride;
    if (pass == 1) {
        // 4 consecutive buckets per thread (one 16-byte load per chunk), block scan of the thread totals
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gbuf) {     // exchange format [F][2048] x (two uint32 counters per word): digits 4k .. 4k+3 are one 16-byte piece
            v = *reinterpret_cast<const uint4 *>(gbuf + static_cast<size_t>(f) * (kBins1 / 2) + 2 * k);
        } else if (p1_u16) {   // partials of k_transpose_count: uint16 [feature][chunk][4096], eight chunks in flight
            const uint16_t *q0 = reinterpret_cast<const uint16_t *>(partial) + static_cast<size_t>(f) * n_chunks * kBins1 + k * 4;
            int c = 0;
            for (; c + 8 <= n_chunks; c += 8) {
                uint2 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const uint2 *>(q0 + static_cast<size_t>(c + u) * kBins1);
#pragma unroll
                for (int u = 0; u < 8; ++u) { v.x += t[u].x & 0xffffu; v.y += t[u].x >> 16; v.z += t[u].y & 0xffffu; v.w += t[u].y >> 16; }
            }
            for (; c < n_chunks; ++c) {
                const uint2 t = *reinterpret_cast<const uint2 *>(q0 + static_cast<size_t>(c) * kBins1);
                v.x += t.x & 0xffffu; v.y += t.x >> 16; v.z += t.y & 0xffffu; v.w += t.y >> 16;
            }
        } else
        for (int c = 0; c < n_chunks; ++c) {
            const uint4 t = *reinterpret_cast<const uint4 *>(p0 + c * pstride + k * 4);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        v.y += v.x; v.z += v.y; v.w += v.z;
        const uint32_t base = block_scan_incl(v.w, scratch) - v.w;
        sums[k * 4 + 0] = base + v.x; sums[k * 4 + 1] = base + v.y; sums[k * 4 + 2] = base + v.z; sums[k * 4 + 3] = base + v.w;
    } else {
        // one wave per slot row: two digits per lane, wave scan
        for (int row = wave; row < n_slots; row += kTgtThreads / 64) {
            uint32_t a = 0, b = 0;
            if (gbuf) {  // exchange format [slot][F][NB / 2] x (digit 2 lane | digit 2 lane + 1 << 32)
                if (2 * lane < NB) {
                    const uint2 t = *reinterpret_cast<const uint2 *>(gbuf + (static_cast<size_t>(row) * F + f) * (NB / 2) + lane);
                    a = t.x; b = t.y;
                }
            } else if (2 * lane < NB)
                for (int c = 0; c < n_chunks; ++c) {
                    const uint2 t = *reinterpret_cast<const uint2 *>(p0 + c * pstride + row * kSlotStride + 2 * lane);
                    a += t.x; b += t.y;
                }
            const uint32_t incl = wave_scan_incl(a + b);
            if (2 * lane < NB) { sums[row * NB + 2 * lane] = incl - b; sums[row * NB + 2 * lane + 1] = incl; }
        }
    }
    __syncthreads();
    int slot = 0, digit = 0;
    uint32_t rank = 0, prefix = 0;
    if (k < B) {
        if (pass == 1) {
            rank = static_cast<uint32_t>(cum[k]);
        } else {
            slot = st.tgt_slot[static_cast<size_t>(f) * B + k];
            rank = st.tgt_rank[static_cast<size_t>(f) * B + k];
            prefix = st.tgt_prefix[static_cast<size_t>(f) * B + k];
        }
        const uint32_t *hs = sums + slot * NB;      // first digit whose inclusive count reaches the rank
        int lo = 0, hi = NB - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (hs[mid] < rank) lo = mid + 1; else hi = mid; }
        digit = lo;
        // last pass: #{keys <= this threshold} = (keys in front of the target's slot = its global rank - its rank inside the slot) + the
        // slot's inclusive count at the found digit.  The class counts of the tree's ROOT follow from these (kern::hist_reduce, root_le).
        if (pass == 1 && k == 0 && le_out) le_out[static_cast<size_t>(gridDim.x) * B + f] = sums[7];   // #{keys <= key(-inf)}, see below
        const uint32_t le_here = static_cast<uint32_t>(static_cast<uint64_t>(cum[k]) - rank + hs[lo]);
        rank -= lo ? hs[lo - 1] : 0u;
        prefix |= static_cast<uint32_t>(digit) << radix_shift(pass);
        // A selected key in the NaN range is raised to -inf's key afterwards (k_keys_to_floats): the count that belongs to the threshold the
        // step will really use is #{keys <= key(-inf)} = all keys whose first digit is <= 0x007 (NaNs are key 0, -inf is 0x007fffff, nothing
        // else lives below 0x00800000), which pass 1 left behind the table.
        if (pass == 4 && le_out) le_out[static_cast<size_t>(f) * B + k] = prefix < 0x007fffffu ? le_out[static_cast<size_t>(gridDim.x) * B + f] : le_here;
    }
    if (pass == 4) {
        if (k < B) thr_keys[static_cast<size_t>(f) * B + k] = prefix;
        return;
    }
    if (k < kMaxTargets) { tslot[k] = k < B ? slot : -1; tdig[k] = digit; }
    __syncthreads();
    // targets are sorted by rank => (slot, digit) is non-decreasing: a target opens a new child slot iff it differs from k-1
    const bool opens = (k < B) && (k == 0 || tslot[k - 1] != slot || tdig[k - 1] != digit);
    const bool new_parent = (k < B) && (k == 0 || tslot[k - 1] != slot);
    const uint32_t incl = block_scan_incl(opens ? 1u : 0u, scratch);
    const int my_slot = static_cast<int>(incl) - 1;        // slot of target k after this pass
    if (k == B - 1) total_slots = incl;
    if (k < B) {
        st.tgt_slot[static_cast<size_t>(f) * B + k] = static_cast<uint16_t>(my_slot);
        st.tgt_rank[static_cast<size_t>(f) * B + k] = rank;
        st.tgt_prefix[static_cast<size_t>(f) * B + k] = prefix;
    }
    if (pass == 1) {
        uint16_t *m = st.map1 + static_cast<size_t>(f) * kBins1;
        for (int i = threadIdx.x; i < kBins1; i += kTgtThreads) m[i] = kNone;
        __syncthreads();
        if (opens) m[digit] = static_cast<uint16_t>(my_slot);
    } else {
        // children of parent slot s are the new slots off[s] .. off[s+1]-1 in digit order (every parent holds >= 1 target)
        uint16_t *off = st.child_off + (static_cast<size_t>(f) * 2 + (pass - 2)) * 258;
        unsigned long long *bits = reinterpret_cast<unsigned long long *>(st.child_bits) + (static_cast<size_t>(f) * 2 + (pass - 2)) * 512;
        for (int i = threadIdx.x; i < 2 * n_slots; i += kTgtThreads) bits[i] = 0ull;
        __syncthreads();
        if (opens) atomicOr(&bits[slot * 2 + (digit >> 6)], 1ull << (digit & 63));
        if (new_parent) off[slot] = static_cast<uint16_t>(my_slot);
    }
    __syncthreads();
    if (k == 0) st.n_slots[f * 4 + (pass - 1)] = total_slots;
}

// ---- row-sharded runs: the digit counts of a pass are summed over ranks before the target pass ---------------------------
// Exchange format: two uint32 counters per int64 word (global row count < 2^32, so the low field never carries), layout
// pass 1 [F][2048], later passes [slot][F][NB/2] so that the slots in use on any feature form a contiguous prefix.
__global__ __launch_bounds__(256) void k_radix_globalize(int pass, const uint32_t *__restrict__ partial, int n_chunks, int F,
                                                         RadixState st, int64_t *__restrict__ gbuf, int p1_u16) {
    const int NB = radix_bins(pass), NBh = NB / 2;
    const int rows = pass == 1 ? 1 : kMaxTargets;
    const size_t pstride = pass == 1 ? kBins1 : kMaxTargets * kSlotStride;
    const size_t total = static_cast<size_t>(rows) * F * NBh;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int pair = static_cast<int>(i % NBh);
        const int f = static_cast<int>((i / NBh) % F);
        const int slot = static_cast<int>(i / (static_cast<size_t>(NBh) * F));
        uint32_t a = 0, b = 0;
        const int live = pass == 1 ? 1 : static_cast<int>(st.n_slots[f * 4 + (pass - 2)]);
        if (p1_u16 && pass == 1) {
            const uint16_t *p = reinterpret_cast<const uint16_t *>(partial) + static_cast<size_t>(f) * n_chunks * kBins1 + 2 * pair;
            for (int c = 0; c < n_chunks; ++c) { a += p[static_cast<size_t>(c) * kBins1]; b += p[static_cast<size_t>(c) * kBins1 + 1]; }
        } else if (slot < live)
            for (int c = 0; c < n_chunks; ++c) {
                const uint32_t *p = partial + (static_cast<size_t>(f) * n_chunks + c) * pstride + (pass == 1 ? 0 : slot * kSlotStride) + 2 * pair;
                a += p[0]; b += p[1];
            }
        gbuf[i] = static_cast<int64_t>(static_cast<uint64_t>(a) | (static_cast<uint64_t>(b) << 32));
    }
}
constexpr size_t align16(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

}  // namespace

size_t radix_state_bytes(int F, int B) {
    const size_t f = static_cast<size_t>(F);
    return align16(f * kBins1 * 2) + align16(f * 2 * 258 * 2) + align16(f * 2 * 512 * 8) + align16(f * 4 * 4) + align16(f * B * 2) +
           2 * align16(f * B * 4) + align16(f * kChunksN * 4);
}
size_t radix_list_bytes(int n, int F) { return sizeof(uint32_t) * (static_cast<size_t>(F) * (static_cast<size_t>(n) / 4 + 64) + 64); }
size_t radix_partial_bytes(int F) {
    const size_t a = static_cast<size_t>(F) * kChunks1 * kBins1, b = static_cast<size_t>(F) * kChunksN * kMaxTargets * kSlotStride;
    return sizeof(uint32_t) * (a > b ? a : b);
}
int radix_max_targets() { return kMaxTargets; }
size_t radix_exchange_words(int F) { return static_cast<size_t>(kMaxTargets) * F * (kSlotStride / 2); }   // int64 words of gbuf
size_t radix_global_partial_bytes(int F) { return sizeof(uint32_t) * static_cast<size_t>(F) * kMaxTargets * kSlotStride; }

// Exact order statistics of every column: thr_keys[f][k] = key of 1-based rank cum[k] in column f of kt ([F][n] ordered keys).
// cum must be non-decreasing, 1 <= cum[k] <= n, B <= 256.
int radix_select(const uint32_t *kt, int n, int F, const int64_t *cum, int B, void *state, uint32_t *partial, uint32_t *lists,
                 uint32_t *thr_keys, hipStream_t s, const RadixComm *comm, int pass1_chunks, uint32_t *le_out) {
    char *p = static_cast<char *>(state);
    auto take = [&](size_t bytes) { char *q = p; p += align16(bytes); return q; };
    const size_t f = static_cast<size_t>(F);
    RadixState st;
    st.map1 = reinterpret_cast<uint16_t *>(take(f * kBins1 * 2));
    st.child_off = reinterpret_cast<uint16_t *>(take(f * 2 * 258 * 2));
    st.child_bits = reinterpret_cast<uint64_t *>(take(f * 2 * 512 * 8));
    st.n_slots = reinterpret_cast<uint32_t *>(take(f * 4 * 4));
    st.tgt_slot = reinterpret_cast<uint16_t *>(take(f * B * 2));
    st.tgt_rank = reinterpret_cast<uint32_t *>(take(f * B * 4));
    st.tgt_prefix = reinterpret_cast<uint32_t *>(take(f * B * 4));
    st.list_cnt = reinterpret_cast<uint32_t *>(take(f * kChunksN * 4));
    st.lists = lists;
    const size_t aux = kBins1 * 2 + 2 * 512 * 8 + 2 * 260 * 2 + kFiltWords * 4 + (kRadixThreads / 64) * 128 * 4;
    const size_t lds1 = static_cast<size_t>(kCopies1) * kBins1 * 4 + aux, lds23 = static_cast<size_t>(kMaxTargets) * 128 * 4 + aux,
                 lds4 = static_cast<size_t>(kMaxTargets) * 64 * 4 + aux;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_radix_count<2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds23));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_radix_count<3>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds23));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_radix_count<4>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds4));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_radix_targets), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxTargets * 128 * 4);
    }
    int cn = 1;
    if (n >= (1 << 16)) { cn = 2; while (cn < kChunksN && cn * F < 256) cn *= 2; }
    const int c1 = n >= (1 << 16) ? kChunks1 : 1;
    // a list holds up to a quarter of its chunk (radix_list_bytes); a chunk with more survivors is streamed again by pass 4
    st.list_cap = static_cast<uint32_t>((n / 4) / cn);
    // Row-sharded runs: after every counting pass the per-rank digit counts are summed over ranks (one all-reduce of packed
    // counters per pass) and the target pass -- identical on every rank -- works on the global counts.
    std::vector<uint32_t> h_slots;
    auto finish_pass = [&](int pass, int n_chunks_pass) -> int {
        const uint32_t *src = partial;
        int chunks = n_chunks_pass;
        int p1_u16 = (pass == 1 && pass1_chunks > 0) ? 1 : 0;
        if (comm) {
            int rows = 1;
            hipLaunchKernelGGL(k_radix_globalize, dim3(1024), dim3(256), 0, s, pass, partial, n_chunks_pass, F, st, comm->gbuf, p1_u16);
            p1_u16 = 0;   // the summed counts come back as uint32
            if (comm->stream_ordered) {
                // (round 5) no read-back, no synchronisation: a feature has at most one slot per target, so B slot rows bound every message
                // (rows nobody filled are summed and ignored: k_radix_targets stops at the feature's own slot count)
                if (pass >= 2) rows = std::min(B, kMaxTargets);
            } else {
                if (pass >= 2) {
                    h_slots.resize(static_cast<size_t>(F) * 4);
                    if (hipMemcpyAsync(h_slots.data(), st.n_slots, sizeof(uint32_t) * h_slots.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
                }
                if (hipStreamSynchronize(s) != hipSuccess) return 1;
                if (pass >= 2) {
                    rows = 0;
                    for (int f2 = 0; f2 < F; ++f2) rows = std::max(rows, static_cast<int>(h_slots[f2 * 4 + (pass - 2)]));
                }
            }
            const size_t words = static_cast<size_t>(rows) * F * (radix_bins(pass) / 2);
            if (comm->allreduce_sum_i64(comm->ctx, comm->gbuf, words) != 0) return 2;
            chunks = 1;     // (round 6: the target pass reads the exchanged words in place -- no unpacking launch)
        }
        const size_t lds = pass == 1 ? kBins1 * 4 : static_cast<size_t>(kMaxTargets) * radix_bins(pass) * 4;
        hipLaunchKernelGGL(k_radix_targets, dim3(F), dim3(kTgtThreads), lds, s, pass, src, chunks, cum, B, st, thr_keys, p1_u16, le_out,
                           comm ? comm->gbuf : nullptr, F);
        return 0;
    };
    int rc = 0;
    const bool plain_loads = [] { const char *e = hooks::raw(hooks::RADIX_PLAIN_LOADS); return e && e[0] == '1'; }();   // measurement hook
    if (pass1_chunks > 0) {   // the first digit was counted by transpose_keys_count
        if ((rc = finish_pass(1, pass1_chunks)) != 0) return rc;
    } else {
        hipLaunchKernelGGL(k_radix_count<1>, dim3(c1, F), dim3(kRadixThreads), lds1, s, kt, n, c1, B, st, partial, plain_loads);
        if ((rc = finish_pass(1, c1)) != 0) return rc;
    }
    hipLaunchKernelGGL(k_radix_count<2>, dim3(cn, F), dim3(kRadixThreads), lds23, s, kt, n, cn, B, st, partial, plain_loads);
    if ((rc = finish_pass(2, cn)) != 0) return rc;
    hipLaunchKernelGGL(k_radix_count<3>, dim3(cn, F), dim3(kRadixThreads), lds23, s, kt, n, cn, B, st, partial, plain_loads);
    if ((rc = finish_pass(3, cn)) != 0) return rc;
    hipLaunchKernelGGL(k_radix_count<4>, dim3(cn, F), dim3(kRadixThreads), lds4, s, kt, n, cn, B, st, partial, plain_loads);
    return finish_pass(4, cn);
}

// Transpose + first-digit counts in one pass over the observations.  Returns the number of row chunks of the uint16 partial
// (to be handed to radix_select as pass1_chunks), or 0 when the shape does not fit (the caller then transposes with
// transpose_keys and radix_select counts the first digit itself).
int transpose_keys_count(const float *obs, int n, int F, uint32_t *kt, uint32_t *partial, hipStream_t s) {
    const char *env_off = hooks::raw(hooks::TRANSPOSE_COUNT);   // measurement / test hook, read per call
    const bool off = env_off && env_off[0] == '0';
    const int chunks = (n + kTcChunk - 1) / kTcChunk;
    if (off || n < (1 << 16) || (n & 3) || (F & 3) || chunks > kTcMaxChunks || (reinterpret_cast<uintptr_t>(obs) & 15) ||
        (reinterpret_cast<uintptr_t>(kt) & 15))
        return 0;
    const size_t lds = sizeof(uint32_t) * (16 * (kBins1 / 2) + kTcWaves * 16 * kTcPatch);
    // The kernel needs 148 KiB of dynamic LDS.  A device that refuses the opt-in (or the launch) is remembered: 0 = "nothing was
    // done", the caller falls back to transpose_keys + a separate first counting pass.
    static PerDeviceOnce attr;
    static uint64_t unsupported = 0;   // per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0;
    if (attr.first() && hipFuncSetAttribute(reinterpret_cast<const void *>(k_transpose_count), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) {
        (void)hipGetLastError();
        unsupported |= bit;
    }
    if (unsupported & bit) return 0;
    hipLaunchKernelGGL(k_transpose_count, dim3(chunks, (F + 15) / 16), dim3(kTcThreads), lds, s, obs, n, F, kt,
                       reinterpret_cast<uint16_t *>(partial), chunks);
    if (hipGetLastError() != hipSuccess) { unsupported |= bit; return 0; }
    return chunks;
}

}  // namespace kern
}  // namespace gbrl
