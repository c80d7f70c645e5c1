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
__host__ __device__ inline int near_core_words(int D, int tile_floats = kNearTile) { return ((2 * D + 3) & ~3) + tile_floats + kNearTileRows / 2 + 4; }

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
                                  int tile_floats = kNearTile /* a multiple of 4, at least 4 * ((D + 3) & ~3) */) {
#pragma clang fp contract(off)
    const int D = a.D, n_l = n - n_r;
    const int n_threads = blockDim.x, half = n_threads / 2;
    float *mean = reinterpret_cast<float *>(scratch);                 // [2][D] right | left
    float *tile = mean + ((2 * D + 3) & ~3);                          // [tile_floats] gradients / products of a batch of rows (16-byte aligned: read as float4)
    uint16_t *tpos = reinterpret_cast<uint16_t *>(tile + tile_floats);  // [kNearTileRows] a row's place among the batch's rows of its side | side << 15
    int *s_nrb_p = reinterpret_cast<int *>(tpos + kNearTileRows);     // rows of the batch that go right
    float *s_num = reinterpret_cast<float *>(s_nrb_p + 1);            // [2]
    const float nrf = static_cast<float>(n_r), nlf = static_cast<float>(n_l);
    const float rrec = n_r > 0 ? 1.0f / nrf : 0.0f, lrec = n_l > 0 ? 1.0f / nlf : 0.0f;
    // A batch of rows is staged in LDS COMPACTED BY SIDE -- the right rows' entries first, then the left rows', each in ascending row order, a
    // row every Dp = D rounded up to 4 floats -- so that the serial loops below walk contiguous memory with nothing to decide per row (a flag
    // test and a dependent LDS read per row cost ~100 clocks each; the first version spent 0.8 ms on a 4096-row node that way).
    const int Dp = (D + 3) & ~3, D4 = D & ~3;
    const int rows_per = max(1, min(kNearTileRows, tile_floats / Dp));
    auto stage_places = [&](int r0, int nr) {     // wave 0: the places (ballot ranks), and how many of the batch's rows go right
        if (threadIdx.x < kWave) {
            int base_r = 0, base_l = 0;
            for (int q0 = 0; q0 < nr; q0 += kWave) {
                const int r = q0 + static_cast<int>(threadIdx.x);
                const bool in = r < nr;
                const bool right = in && (static_cast<uint32_t>(ent[r0 + min(r, nr - 1)]) >> 31) != 0;
                const unsigned long long mr = __ballot(right), ml = __ballot(in && !right);
                const unsigned long long below = threadIdx.x == 0 ? 0ull : (~0ull >> (kWave - threadIdx.x));
                if (in) tpos[r] = static_cast<uint16_t>(right ? (0x8000 | (base_r + __popcll(mr & below))) : (base_l + __popcll(ml & below)));
                base_r += __popcll(mr);
                base_l += __popcll(ml);
            }
            if (threadIdx.x == 0) *s_nrb_p = base_r;
        }
    };
    // ---- pass 1: per side and column, the float32 sum over the side's rows in ascending order
    for (int c0 = 0; c0 < D; c0 += half) {   // (one pass unless there are more than 128 columns)
        const int side = threadIdx.x / (half);          // 0 right, 1 left
        const int c = c0 + static_cast<int>(threadIdx.x) % (half);
        float sum = 0.0f;
        for (int r0 = 0; r0 < n; r0 += rows_per) {
            const int nr = min(rows_per, n - r0);
            __syncthreads();
            stage_places(r0, nr);
            __syncthreads();
            const int nrb = *s_nrb_p;
            for (int e = threadIdx.x; e < nr * D; e += n_threads) {
                const int r = e / D, cc = e - r * D;
                const unsigned tp = tpos[r];
                tile[((tp & 0x8000u) ? (tp & 0x7fffu) : (nrb + tp)) * Dp + cc] = near_grad(a, ent[r0 + r] & 0x7fffffff, cc);
            }
            __syncthreads();
            if (c < D) {
                const float *col = tile + (side ? nrb * Dp : 0) + c;
                const int m = side ? nr - nrb : nrb;
#pragma unroll 8
                for (int r = 0; r < m; ++r) sum += col[r * Dp];
            }
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
            for (int e = threadIdx.x; e < nr * D; e += n_threads) {
                const int r = e / D, cc = e - r * D;
                const unsigned tp = tpos[r];
                const bool right = (tp & 0x8000u) != 0;
                const float g = near_grad(a, ent[r0 + r] & 0x7fffffff, cc);
                // (the vectorised columns' products are rounded here, before the add; the last D % 4 columns are fused below)
                tile[(right ? (tp & 0x7fffu) : (nrb + tp)) * Dp + cc] = cc < D4 ? g * mean[(right ? 0 : D) + cc] : g;
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
