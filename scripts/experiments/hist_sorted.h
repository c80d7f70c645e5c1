// hist_sorted.h -- split-score histogram build WITHOUT one LDS atomic per (row, feature, output).
//
// k_hist_build (kernels.hip) spends D + 1 = 9 `ds_add_u32` per (row, feature): the LDS write path moves 64 B/clk/CU, one
// wave-instruction per ~4.5 clk, which caps it at 0.50 ms per depth-6 tree at config 2.  This kernel replaces
// the scatter by a gather: per sub-chunk of kSortRows rows of the block's chunk and per feature of the block's group,
//   1. a counting sort of the rows by class in LDS: ONE returning `ds_add_rtn_u32` per (row, feature) on the class counter
//      (its return value is the row's rank inside the class), an exclusive scan of the counters, and one `ds_write_b16` that
//      puts the row's gradient-record offset at start[class] + rank;
//   2. a register accumulation: wave f owns feature f, lane l owns classes l, l + 64, l + 128, ...; it walks the rows of each of
//      its classes (contiguous in the sorted list), reads their 16-byte gradient records (8 x int16, staged once per sub-chunk and
//      shared by the 16 features) with `ds_read_b128` -- the LDS read path moves 256 B/clk/CU -- and adds them into int32 registers
//      that live across all sub-chunks of the block.
// Integer sums of the same int32 values: the partials are bit-identical to k_hist_build's, in the same layout
// [chunk][group][class][D+1][16], so everything downstream is unchanged.
//
// Scope: 16 features per block, D <= 8 (the gradient record holds 8 x int16: needs max|q| < 2^15, which the fixed-point scale
// guarantees whenever the block's row cap is >= 65536 -- StepScales; the launcher checks), classes <= 64 * kSortSlots.
#pragma once

#include "../../gbrl_amd/csrc/kernels.h"

namespace gbrl {
namespace kern {
namespace {

constexpr int kSortRows = 2048;        // rows per sub-chunk: 32 KiB of gradient records + 64 KiB of sorted lists + 20 KiB counters
constexpr int kSortSlots = 5;          // classes per lane (257 classes = 4 x 64 + 1)
constexpr int kSortNBP = 64 * kSortSlots;
constexpr int kSortThreads = 1024;

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = __lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

typedef int v4i_t __attribute__((ext_vector_type(4)));
#ifdef HS_PROFILE
__device__ long long hs_profile[8];
#endif

template <int DT>
__global__ __launch_bounds__(kSortThreads) void k_hist_sorted(const uint16_t *__restrict__ codes, int n_rows,
                                                               const int32_t *__restrict__ qg, const int32_t *__restrict__ rows,
                                                               const Chunk *__restrict__ chunks, int n_chunks, int n_groups, int NB,
                                                               int32_t *__restrict__ partials) {
    extern __shared__ int32_t h[];
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int g = jj % n_groups;
    const int chunk_id = (jj / n_groups) * 8 + xcd;
    if (chunk_id >= n_chunks) return;
    const Chunk ck = chunks[chunk_id];
    if (ck.len <= 0) return;
    v4i_t *G = reinterpret_cast<v4i_t *>(h);                                               // [kSortRows] 8 x int16
    uint16_t *sorted = reinterpret_cast<uint16_t *>(h + kSortRows * 4);                     // [16][kSortRows] byte offsets into G
    uint32_t *cnt = reinterpret_cast<uint32_t *>(sorted + 16 * kSortRows);                  // [16][kSortNBP] counters, then class starts
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int32_t *rlist = rows + ck.start;
    const uint16_t *cgroup = codes + static_cast<size_t>(g) * n_rows * kCodeGroup;
    int acc[kSortSlots][8];
    int cacc[kSortSlots];
#pragma unroll
    for (int j = 0; j < kSortSlots; ++j) {
        cacc[j] = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[j][d] = 0;
    }
    constexpr int RPT = kSortRows / kSortThreads;   // rows per thread and sub-chunk
#ifdef HS_PROFILE
    long long t_ph[6] = {0, 0, 0, 0, 0, 0}, t_last = clock64();
#define HS_MARK(k) { const long long t_now = clock64(); t_ph[k] += t_now - t_last; t_last = t_now; }
#else
#define HS_MARK(k)
#endif
    for (int sub = 0; sub < ck.len; sub += kSortRows) {
        const int m = min(kSortRows, ck.len - sub);
        for (int i = tid; i < 16 * kSortNBP; i += kSortThreads) cnt[i] = 0;
        __syncthreads();
        HS_MARK(0)
        // ---- phase 1: stage the gradient records, rank every (row, feature) inside its class ----
        uint32_t cw[RPT][8];       // the row's 16 class codes, two per word
        uint32_t rk[RPT][8];       // the 16 ranks, two per word
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int p = tid + u * kSortThreads;
            if (p < m) {
                const int row = rlist[sub + p];
                const uint4 *cp = reinterpret_cast<const uint4 *>(cgroup + static_cast<size_t>(row) * kCodeGroup);
                const uint4 ca = cp[0], cb = cp[1];
                cw[u][0] = ca.x; cw[u][1] = ca.y; cw[u][2] = ca.z; cw[u][3] = ca.w;
                cw[u][4] = cb.x; cw[u][5] = cb.y; cw[u][6] = cb.z; cw[u][7] = cb.w;
                v4i_t rec;
                if (DT == 8) {
                    const int4 *qp = reinterpret_cast<const int4 *>(qg + static_cast<size_t>(row) * 8);
                    const int4 a = qp[0], b = qp[1];
                    rec.x = (a.x & 0xffff) | (a.y << 16); rec.y = (a.z & 0xffff) | (a.w << 16);
                    rec.z = (b.x & 0xffff) | (b.y << 16); rec.w = (b.z & 0xffff) | (b.w << 16);
                } else {
                    int q[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) q[d] = d < DT ? qg[static_cast<size_t>(row) * DT + d] : 0;
                    rec.x = (q[0] & 0xffff) | (q[1] << 16); rec.y = (q[2] & 0xffff) | (q[3] << 16);
                    rec.z = (q[4] & 0xffff) | (q[5] << 16); rec.w = (q[6] & 0xffff) | (q[7] << 16);
                }
                G[p] = rec;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const uint32_t r0 = atomicAdd(&cnt[(2 * w) * kSortNBP + (cw[u][w] & 0xffffu)], 1u);
                    const uint32_t r1 = atomicAdd(&cnt[(2 * w + 1) * kSortNBP + (cw[u][w] >> 16)], 1u);
                    rk[u][w] = r0 | (r1 << 16);
                }
            }
        }
        __syncthreads();
        HS_MARK(1)
        // ---- phase 2: wave f turns feature f's class counts into class starts (classes in index order) ----
        int n[kSortSlots], st[kSortSlots];
        {
            int base = 0;
#pragma unroll
            for (int j = 0; j < kSortSlots; ++j) {
                const int c = lane + 64 * j;
                n[j] = c < NB ? static_cast<int>(cnt[wave * kSortNBP + c]) : 0;
                const int incl = wave_incl_scan(n[j]);
                st[j] = base + incl - n[j];
                base += __shfl(incl, 63, 64);
            }
        }
#pragma unroll
        for (int j = 0; j < kSortSlots; ++j) cnt[wave * kSortNBP + lane + 64 * j] = static_cast<uint32_t>(st[j]);
        __syncthreads();
        HS_MARK(2)
        // ---- phase 3: scatter the record offsets into the class-sorted lists ----
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int p = tid + u * kSortThreads;
            if (p < m) {
                const uint16_t off = static_cast<uint16_t>(p * 16);
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const uint32_t p0 = cnt[(2 * w) * kSortNBP + (cw[u][w] & 0xffffu)] + (rk[u][w] & 0xffffu);
                    const uint32_t p1 = cnt[(2 * w + 1) * kSortNBP + (cw[u][w] >> 16)] + (rk[u][w] >> 16);
                    sorted[(2 * w) * kSortRows + p0] = off;
                    sorted[(2 * w + 1) * kSortRows + p1] = off;
                }
            }
        }
        __syncthreads();
        HS_MARK(3)
        // ---- phase 4: wave f, lane l: the rows of classes l + 64 j are contiguous in feature f's list ----
        const char *Gb = reinterpret_cast<const char *>(G);
#pragma unroll
        for (int j = 0; j < kSortSlots; ++j) {
            if (64 * j >= NB) break;
            const uint16_t *sp = sorted + wave * kSortRows + st[j];
            const int nj = n[j];
            const int nmax = wave_max(nj);
            cacc[j] += nj;
            // reads past the class run (and, for short classes, past the list) land on other valid entries or are masked to a
            // valid record address; only the adds are predicated
            int o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = sp[k];
            for (int i = 0; i < nmax; i += 4) {
                v4i_t r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const v4i_t *>(Gb + (o[k] & (kSortRows * 16 - 16)));
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = sp[i + 4 + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (i + k < nj) {
                        acc[j][0] += static_cast<int16_t>(r[k].x & 0xffff); acc[j][1] += r[k].x >> 16;
                        acc[j][2] += static_cast<int16_t>(r[k].y & 0xffff); acc[j][3] += r[k].y >> 16;
                        acc[j][4] += static_cast<int16_t>(r[k].z & 0xffff); acc[j][5] += r[k].z >> 16;
                        acc[j][6] += static_cast<int16_t>(r[k].w & 0xffff); acc[j][7] += r[k].w >> 16;
                    }
                }
            }
        }
        __syncthreads();
        HS_MARK(4)
    }
#ifdef HS_PROFILE
    if (tid == 0 && blockIdx.x == 0) { for (int k = 0; k < 5; ++k) hs_profile[k] = t_ph[k]; }
#endif
    // ---- the block's sums leave in k_hist_build's layout h[(class * (D + 1) + d) * 16 + feature] ----
#pragma unroll
    for (int j = 0; j < kSortSlots; ++j) {
        const int c = lane + 64 * j;
        if (c < NB) {
#pragma unroll
            for (int d = 0; d < DT; ++d) h[(c * (DT + 1) + d) * 16 + wave] = acc[j][d];
            h[(c * (DT + 1) + DT) * 16 + wave] = cacc[j];
        }
    }
    __syncthreads();
    const int n_acc = NB * (DT + 1) * 16;
    int32_t *out = partials + (static_cast<size_t>(chunk_id) * n_groups + g) * n_acc;
    const int4 *h4 = reinterpret_cast<const int4 *>(h);
    int4 *o4 = reinterpret_cast<int4 *>(out);
    for (int i = tid; i < n_acc / 4; i += kSortThreads) o4[i] = h4[i];
}

inline size_t hist_sorted_lds_bytes(int NB, int D) {
    const size_t work = static_cast<size_t>(kSortRows) * 16 + static_cast<size_t>(16) * kSortRows * 2 + static_cast<size_t>(16) * kSortNBP * 4;
    const size_t out = static_cast<size_t>(NB) * (D + 1) * 16 * 4;
    return work > out ? work : out;
}

}  // namespace
}  // namespace kern
}  // namespace gbrl
