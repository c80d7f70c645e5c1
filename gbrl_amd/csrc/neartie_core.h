// neartie_core.h -- the reference's float32 split score of ONE partition of ONE node, evaluated by a whole block (neartie.hip's replay kernel
// and the one-launch growth kernel of small_grow.hip, which replays its near-ties itself).  The operation sequence is stated at the head of
// neartie.hip and pinned by tests/test_oracle.py / tests/test_gpu_neartie.py; no floating-point contraction in here.
#pragma once

#include "kernels_common.h"

namespace gbrl {
namespace kern {
namespace {

constexpr int kNearTile = 4096, kNearTileRows = 1024;
// 32-bit words of LDS scratch near_replay_core needs (16-byte aligned)
__host__ __device__ inline int near_core_words(int D, int tile_floats = kNearTile, int tile_rows = kNearTileRows) { return ((2 * D + 3) & ~3) + tile_floats + tile_rows / 2 + 4 + (tile_rows + 3) / 4 + tile_rows + ((tile_rows + kWave - 1) / kWave + 1); }
// batches above 65 536 rows (round 6): tiles of up to 3072 rows (96 KiB of staged floats) -- a tile costs two barriers, a staging pass and a serial loop, and a 2^20-row node has thousands of 512-row ones
constexpr int kNearBigTile = 24576, kNearBigTileRows = 3072;

struct NearGrads { const float *grads; const float *meanden; int D; };
__device__ __forceinline__ float near_grad(const NearGrads &a, int row, int c) {
#pragma clang fp contract(off)
    const float g = a.grads[static_cast<size_t>(row) * a.D + c];
    if (a.meanden == nullptr) return g;
    return (g - a.meanden[c]) / a.meanden[a.D + c];     // fitter.cpp:58-63 -> math_ops.cpp:498,94: what k_quantize standardises
}
__device__ __forceinline__ float near_sqnorm(const float *v, int D) {   // math_ops.h:476-485 as compiled
#pragma clang fp contract(off)
    const int D4 = D & ~3;
    float s = 0.0f;
    for (int c = 0; c < D4; ++c) { const float p = v[c] * v[c]; s = s + p; }
    for (int c = D4; c < D; ++c) s = fmaf(v[c], v[c], s);
    return s;
}

// ent[0 .. n): the node's rows in ascending order, bit 31 = the row goes right; n_r of them do.  is_parent: the node's parent score (every row
// on the "left").  Every thread of the block calls it (blockDim.x a multiple of 128, at least 128); the result is valid in thread 0.
// Two passes over the rows, a tile of kNearTile floats at a time, staged in LDS by all threads (one memory round trip per tile): pass 1 the
// per-side column sums (one thread per (side, column) adds its column's entries in order), pass 2 -- Cosine -- the two in-order dot chains, one
// lane each (waves 0 and 1), over products the staging has already rounded.
__device__ float near_replay_core(const int32_t *__restrict__ ent, int n, int n_r, const NearGrads a, bool cosine, bool is_parent, uint32_t *scratch,
                                  int tile_floats = kNearTile /* a multiple of 4, at least 4 * ((D + 3) & ~3) */, int tile_rows = kNearTileRows /* even, at most 32 768 */) {
#pragma clang fp contract(off)
    const int D = a.D, n_l = n - n_r;
    const int n_threads = blockDim.x, half = n_threads / 2;
    float *mean = reinterpret_cast<float *>(scratch);                 // [2][D] right | left
    float *tile = mean + ((2 * D + 3) & ~3);                          // [tile_floats] gradients / products of a batch of rows (16-byte aligned: read as float4)
    uint16_t *tpos = reinterpret_cast<uint16_t *>(tile + tile_floats);  // [kNearTileRows] a row's place among the batch's rows of its side | side << 15
    int *s_nrb_p = reinterpret_cast<int *>(tpos + tile_rows);         // rows of the batch that go right
    float *s_num = reinterpret_cast<float *>(s_nrb_p + 1);            // [2]
    const float nrf = static_cast<float>(n_r), nlf = static_cast<float>(n_l);
    const float rrec = n_r > 0 ? 1.0f / nrf : 0.0f, lrec = n_l > 0 ? 1.0f / nlf : 0.0f;
    // A batch of rows is staged in LDS COMPACTED BY SIDE -- the right rows' entries first, then the left rows', each in ascending row order, a
    // row every Dp = D rounded up to 4 floats -- so that the serial loops below walk contiguous memory with nothing to decide per row (a flag
    // test and a dependent LDS read per row cost ~100 clocks each; the first version spent 0.8 ms on a 4096-row node that way).
    const int Dp = (D + 3) & ~3, D4 = D & ~3;
    const int rows_per = max(1, min(tile_rows, tile_floats / Dp));
    uint8_t *rflag = reinterpret_cast<uint8_t *>(s_num + 2);          // [tile_rows] the batch's side flags, fetched by ALL threads (round 6: wave 0 used to
                                                                      // load them itself, one dependent global round trip per 64 rows -- half of a big node's replay)
    int32_t *trow = reinterpret_cast<int32_t *>(rflag + ((tile_rows + 3) & ~3));   // [tile_rows] the batch's row indices (so that the staging below has ONE global round trip per element)
    auto stage_places = [&](int r0, int nr) {     // wave 0: the places (ballot ranks), and how many of the batch's rows go right
        for (int rb = threadIdx.x; rb < nr; rb += n_threads * 4) {       // (four list entries per thread in flight)
            int32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ent[r0 + min(rb + u * n_threads, nr - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rb + u * n_threads;
                if (r < nr) { rflag[r] = static_cast<uint8_t>(static_cast<uint32_t>(v[u]) >> 31); trow[r] = v[u] & 0x7fffffff; }
            }
        }
        __syncthreads();
        // every wave ranks its own 64-row groups (group g -> wave g % n_waves); the groups' bases come from a prefix over their counts
        const int n_groups = (nr + kWave - 1) / kWave, n_waves = n_threads / kWave, wv = threadIdx.x / kWave, ln = threadIdx.x & (kWave - 1);
        uint16_t *gcnt = reinterpret_cast<uint16_t *>(trow + tile_rows);   // [groups] rows of the group that go right
        uint16_t *gbase = gcnt + ((tile_rows + kWave - 1) / kWave);         // [groups] ... of all groups before it
        for (int g = wv; g < n_groups; g += n_waves) {
            const int r = g * kWave + ln;
            const unsigned long long mr = __ballot(r < nr && rflag[min(r, nr - 1)] != 0);
            if (ln == 0) gcnt[g] = static_cast<uint16_t>(__popcll(mr));
        }
        __syncthreads();
        if (threadIdx.x < kWave) {      // exclusive prefix over the groups' counts: one wave, 64 groups per step
            int carry = 0;
            for (int g0 = 0; g0 < n_groups; g0 += kWave) {
                const int g = g0 + static_cast<int>(threadIdx.x);
                const int c = g < n_groups ? gcnt[g] : 0;
                int inc = c;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(inc, o, kWave); if (static_cast<int>(threadIdx.x) >= o) inc += up; }
                if (g < n_groups) gbase[g] = static_cast<uint16_t>(carry + inc - c);
                carry += __shfl(inc, kWave - 1, kWave);
            }
            if (threadIdx.x == 0) *s_nrb_p = carry;
        }
        __syncthreads();
        for (int g = wv; g < n_groups; g += n_waves) {
            const int r = g * kWave + ln;
            const bool in = r < nr;
            const bool right = in && rflag[min(r, nr - 1)] != 0;
            const unsigned long long mr = __ballot(right), ml = __ballot(in && !right);
            const unsigned long long below = ln == 0 ? 0ull : (~0ull >> (kWave - ln));
            const int base_r = gbase[g], base_l = g * kWave - base_r;      // rows before the group that go right / left
            if (in) tpos[r] = static_cast<uint16_t>(right ? (0x8000 | (base_r + __popcll(mr & below))) : (base_l + __popcll(ml & below)));
        }
    };
#ifdef GBRL_NEAR_PROF
    long long pf[6] = {0, 0, 0, 0, 0, 0}; long long pt0 = clock64();
#define NEAR_PROF(i) { const long long t_ = clock64(); pf[i] += t_ - pt0; pt0 = t_; }
#else
#define NEAR_PROF(i)
#endif
    // ---- pass 1: per side and column, the float32 sum over the side's rows in ascending order.  Round 6 (nodes of up to 2^20 rows): the tile is
    // staged COLUMN-MAJOR -- tile[col * cap + place], the left side's places starting at a multiple of 4 -- so that the one thread that owns a
    // (side, column) chain reads four rows per LDS instruction, and the gradients are fetched a float4 at a time, eight per thread in flight
    // (a 2^20-row node spent 19 ms per pass waiting for 4-byte gathers, one round trip per element, on the four waves of its block).
    typedef float near_f4 __attribute__((ext_vector_type(4)));
    // column-major capacity: right rows at [0, nrb), left rows from the next multiple of 4 -> a column needs up to rows + 3 places, rounded to 4
    const int cap_max = (tile_floats / D) & ~3;                       // (>= 4: tile_floats >= 4 * Dp)
    const int rows_p1 = max(1, min(tile_rows, cap_max - 3));
    const int cap = (rows_p1 + 6) & ~3;
    const bool vec4 = D4 == D && D >= 4 && (reinterpret_cast<uintptr_t>(a.grads) & 15) == 0;
    for (int c0 = 0; c0 < D; c0 += half) {   // (one pass unless there are more than 128 columns)
        const int side = threadIdx.x / (half);          // 0 right, 1 left
        const int c = c0 + static_cast<int>(threadIdx.x) % (half);
        float sum = 0.0f;
        for (int r0 = 0; r0 < n; r0 += rows_p1) {
            const int nr = min(rows_p1, n - r0);
            __syncthreads();
            NEAR_PROF(5)
            stage_places(r0, nr);
            __syncthreads();
            NEAR_PROF(0)
            const int nrb = *s_nrb_p, nrbA = (nrb + 3) & ~3;
            if (vec4) {
                const int Q = D >> 2;
                for (int u0 = threadIdx.x; u0 < nr * Q; u0 += n_threads * 8) {
                    int pl[8], q4[8]; near_f4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int uu = min(u0 + u * n_threads, nr * Q - 1);
                        const int r = uu / Q;
                        q4[u] = (uu - r * Q) * 4;
                        const unsigned tp = tpos[r];
                        pl[u] = (tp & 0x8000u) ? static_cast<int>(tp & 0x7fffu) : (nrbA + static_cast<int>(tp));
                        v[u] = *reinterpret_cast<const near_f4 *>(a.grads + static_cast<size_t>(trow[r]) * D + q4[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u0 + u * n_threads >= nr * Q) continue;
                        float g4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int cc = q4[u] + k;
                            const float g = a.meanden == nullptr ? g4[k] : (g4[k] - a.meanden[cc]) / a.meanden[D + cc];   // near_grad's operations
                            tile[cc * cap + pl[u]] = g;
                        }
                    }
                }
            } else {
                for (int e0 = threadIdx.x; e0 < nr * D; e0 += n_threads * 8) {
                    int dst[8]; float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = min(e0 + u * n_threads, nr * D - 1);
                        const int r = e / D, cc = e - r * D;
                        const unsigned tp = tpos[r];
                        dst[u] = cc * cap + ((tp & 0x8000u) ? static_cast<int>(tp & 0x7fffu) : (nrbA + static_cast<int>(tp)));
                        g[u] = near_grad(a, trow[r], cc);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (e0 + u * n_threads < nr * D) tile[dst[u]] = g[u];
                }
            }
            __syncthreads();
            NEAR_PROF(1)
            if (c < D) {
                const float *col = tile + c * cap + (side ? nrbA : 0);   // 16-byte aligned
                const int m = side ? nr - nrb : nrb;
                const near_f4 *col4 = reinterpret_cast<const near_f4 *>(col);
                const int m4 = m >> 2;
                // the adds are one dependent chain; the LDS reads of the NEXT eight float4s are issued before the current eight are added
                int q = 0;
                if (m4 >= 8) {
                    near_f4 cur[8], nxt[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cur[u] = col4[u];
                    for (q = 8; q + 8 <= m4; q += 8) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) nxt[u] = col4[q + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { sum = sum + cur[u].x; sum = sum + cur[u].y; sum = sum + cur[u].z; sum = sum + cur[u].w; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sum = sum + cur[u].x; sum = sum + cur[u].y; sum = sum + cur[u].z; sum = sum + cur[u].w; }
                }
                for (; q < m4; ++q) { const near_f4 v = col4[q]; sum = sum + v.x; sum = sum + v.y; sum = sum + v.z; sum = sum + v.w; }
                for (int r = m4 * 4; r < m; ++r) sum = sum + col[r];
            }
            NEAR_PROF(2)
        }
        if (c < D) mean[side * D + c] = sum * (side ? lrec : rrec);
    }
    __syncthreads();
    // ---- pass 2 (Cosine): sum_{row, col} g[row][col] * mean_side[col] per side, in order (math_ops.h:432-449 as compiled: the head of this file)
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    if (cosine) {
        float s = 0.0f;    // lane 0 of wave 0: the right side's chain; of wave 1: the left side's
        for (int r0 = 0; r0 < n; r0 += rows_per) {
            const int nr = min(rows_per, n - r0);
            __syncthreads();
            stage_places(r0, nr);
            __syncthreads();
            const int nrb = *s_nrb_p;
            if (vec4) {
                const int Q = D >> 2;
                for (int u0 = threadIdx.x; u0 < nr * Q; u0 += n_threads * 8) {
                    int dst[8], q4[8], sd[8]; near_f4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int uu = min(u0 + u * n_threads, nr * Q - 1);
                        const int r = uu / Q;
                        q4[u] = (uu - r * Q) * 4;
                        const unsigned tp = tpos[r];
                        const bool right = (tp & 0x8000u) != 0;
                        sd[u] = right ? 0 : D;
                        dst[u] = (right ? static_cast<int>(tp & 0x7fffu) : (nrb + static_cast<int>(tp))) * Dp + q4[u];
                        v[u] = *reinterpret_cast<const near_f4 *>(a.grads + static_cast<size_t>(trow[r]) * D + q4[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u0 + u * n_threads >= nr * Q) continue;
                        float g4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        near_f4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int cc = q4[u] + k;
                            const float g = a.meanden == nullptr ? g4[k] : (g4[k] - a.meanden[cc]) / a.meanden[D + cc];   // near_grad's operations
                            const float pr = g * mean[sd[u] + cc];          // (rounded here, before the add)
                            if (k == 0) o.x = pr; else if (k == 1) o.y = pr; else if (k == 2) o.z = pr; else o.w = pr;
                        }
                        *reinterpret_cast<near_f4 *>(tile + dst[u]) = o;
                    }
                }
            } else {
            for (int e0 = threadIdx.x; e0 < nr * D; e0 += n_threads * 8) {
                int dst[8]; float g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * n_threads, nr * D - 1);
                    const int r = e / D, cc = e - r * D;
                    const unsigned tp = tpos[r];
                    const bool right = (tp & 0x8000u) != 0;
                    dst[u] = (right ? (tp & 0x7fffu) : (nrb + tp)) * Dp + cc;
                    const float gv = near_grad(a, trow[r], cc);
                    // (the vectorised columns' products are rounded here, before the add; the last D % 4 columns are fused below)
                    g[u] = cc < D4 ? gv * mean[(right ? 0 : D) + cc] : gv;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (e0 + u * n_threads < nr * D) tile[dst[u]] = g[u];
            }
            }
            __syncthreads();
            if (wave < 2 && lane == 0) {
                const float *b = tile + (wave ? nrb * Dp : 0);
                const float *vec = mean + (wave ? D : 0);
                const int m = wave ? nr - nrb : nrb;
                typedef float near_f4 __attribute__((ext_vector_type(4)));
                if (D4 == D) {          // whole rows of rounded products: one contiguous in-order sum
                    const near_f4 *b4 = reinterpret_cast<const near_f4 *>(b);
                    const int n4 = m * (D / 4);
#pragma unroll 8
                    for (int q = 0; q < n4; ++q) { const near_f4 v = b4[q]; s = s + v.x; s = s + v.y; s = s + v.z; s = s + v.w; }
                } else {
#pragma unroll 4
                    for (int r = 0; r < m; ++r) {
                        const float *br = b + r * Dp;
                        for (int c = 0; c < D4; c += 4) { const near_f4 v = *reinterpret_cast<const near_f4 *>(br + c); s = s + v.x; s = s + v.y; s = s + v.z; s = s + v.w; }
                        for (int c = D4; c < D; ++c) s = fmaf(br[c], vec[c], s);
                    }
                }
            }
        }
        if (wave < 2 && lane == 0) s_num[wave] = s;
    }
    __syncthreads();
#ifdef GBRL_NEAR_PROF
    NEAR_PROF(3)
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)
        printf("[near prof] n %d  places %.0f us  staging %.0f us  serial %.0f us  pass2 %.0f us  barrier-wait %.0f us\n", n, pf[0] / 100.0, pf[1] / 100.0, pf[2] / 100.0, pf[3] / 100.0, pf[5] / 100.0);
#endif
    float res = 0.0f;
    if (threadIdx.x == 0) {
        if (is_parent) {       // every row is on the "left" side here
            if (cosine) {
                const float den = near_sqnorm(mean + D, D) * nlf;
                res = (n_l == 0 || den == 0.0f) ? 0.0f : static_cast<float>(static_cast<double>(s_num[1]) / sqrt(static_cast<double>(den)));
            } else {
                res = near_sqnorm(mean + D, D) * nlf;
            }
        } else if (cosine) {
            const float tn = near_sqnorm(mean, D), fn = near_sqnorm(mean + D, D);
            const float fden = fn * nlf;
            const float den = fmaf(tn, nrf, fden);
            const float num = s_num[0] + s_num[1];
            res = den == 0.0f ? 0.0f : num / sqrtf(den);
        } else {
            const float ln = near_sqnorm(mean + D, D), rn = near_sqnorm(mean, D);
            const float rp = nrf * rn;
            res = fmaf(nlf, ln, rp);
        }
    }
    return res;
}

}  // namespace
}  // namespace kern
}  // namespace gbrl
