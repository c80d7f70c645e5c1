// score_common.h -- split-score arithmetic shared by k_score (kernels.hip) and the one-launch RL-sized growth kernel (small_grow.hip):
// both evaluate a candidate from exact integer sums with the SAME fp64 expression, so a step grows the same bytes whichever path it takes.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "kernels_common.h"

namespace gbrl {
namespace kern {
namespace {

template <typename T>
__device__ __forceinline__ double side_term(const T *S, int D, int64_t n, double inv_scale) {
    if (n <= 0) return 0.0;
    double ss = 0.0;
    for (int d = 0; d < D; ++d) {
        const double v = static_cast<double>(S[d]) * inv_scale;
        ss += v * v;
    }
    return ss / static_cast<double>(n);
}

// A8 oblivious: s_j = (sum over nodes, in node order, fp32) * w_j ; lowest index among maxima wins (fitter.cpp:411-457)
struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) {
    // strictly greater wins; on equality the lower reference index wins.  A -inf score never replaces the initial
    // (-inf, none) state, exactly like "if (score > local_best_score)" with local_best = -inf (fitter.cpp:338, 441).
    if (b.v > a.v || (b.v == a.v && b.v > -INFINITY && b.i < a.i)) return b;
    return a;
}
// Near-tie detection: (best, second) pairs, `second` = the best value STRICTLY below `best` (candidates with equal exact gains split the
// node's rows identically -- thresholds between the same two rows -- and are not a tie to resolve).  Returns the merged pair's second.
__device__ __forceinline__ float second_distinct(float a1, float a2, float c1, float c2) {
    const float hi = fmaxf(a1, c1), lo = fminf(a1, c1);
    float s = fmaxf(a2, c2);
    if (lo < hi) s = fmaxf(s, lo);
    return s;
}
// The same with the candidates' identity taken from (gain, rows going right): two candidates with the SAME float32 gain and the same child
// sizes split the node's rows identically (thresholds between the same two rows); the same gain with different child sizes is a tie between
// DIFFERENT partitions -- mirrored ones, or scores closer than float32 resolves -- and counts as a runner-up at distance 0.
__device__ __forceinline__ float second_merge(float v1, int nr1, float s1, float v2, int nr2, float s2) {
    float s = fmaxf(s1, s2);
    if (v1 == v2) { if (nr1 != nr2 && v1 > -INFINITY) s = v1; }
    else s = fmaxf(s, fminf(v1, v2));
    return s;
}
// The class of a candidate that sends n_right of its node's n_node rows right.  Every row on one side -- either side -- is ONE class (the
// reference scores both the same, operation for operation).  Nodes of up to 1024 rows tell classes apart by the gain alone: mirror-image
// partitions of a few rows tie exactly all the time there (one tree in five flagged at 64 rows, none in 500 at 1024 on random data).  The
// callers apply it to batches of more than 8192 rows only: in the one-launch kernel of the smaller ones carrying the child size through the
// selection costs 3.5 % of every step (profiles/r05_neartie_cost.txt), and both growth paths must flag the same nodes.
constexpr int kNearClassRows = 1024;
__device__ __forceinline__ int near_class(long long n_right, long long n_node) { return (n_node <= kNearClassRows || n_right == n_node) ? 0 : static_cast<int>(n_right); }
__device__ __forceinline__ bool better_takes_second(Best a, Best b) { return b.v > a.v || (b.v == a.v && b.v > -INFINITY && b.i < a.i); }
// The window of the near-tie replay, relative to the scores' magnitude: `rel` (2^-20) for nodes of up to 8192 rows -- the RL-sized range, where
// a replay costs as much as the tree -- and the reference's own summation noise eps32 * sqrt(rows) beyond (tests/neartie.py explains a
// difference within 4x that).
__device__ __forceinline__ float near_window_rel(float rel, long long rows) {
    return rows > 8192 ? fmaxf(rel, 1.1920929e-07f * sqrtf(static_cast<float>(rows))) : rel;
}
// Inclusive scan over the 64 lanes of NINE 64-bit values at once (k_score's fields), every step as two DPP-fused adds per value:
//   v_add_co_u32_dpp lo, vcc, lo, lo <ctrl>   ;  v_addc_co_u32_dpp hi, vcc, hi, hi, vcc <ctrl>
// A lane whose source is outside its row (row_shr) or whose row is masked (row_bcast) is simply not written, i.e. keeps x -- no
// zero-initialised temporary, no separate 64-bit add: 2 VALU instructions per value and step instead of 5 (the kernel is bound by its
// VALU instruction count).  gfx9 hazard "VALU writes a VGPR, DPP reads it: 2 wait states": inside a step the nine values are
// independent, and a value is touched again 16 instructions later; the s_nop covers whatever the compiler issued just before.
#define GBRL_SCAN9_STEP(CTRL) \
    asm volatile("s_nop 1\n" \
                 "v_add_co_u32_dpp %0, vcc, %0, %0 " CTRL "\n v_addc_co_u32_dpp %9, vcc, %9, %9, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %1, vcc, %1, %1 " CTRL "\n v_addc_co_u32_dpp %10, vcc, %10, %10, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %2, vcc, %2, %2 " CTRL "\n v_addc_co_u32_dpp %11, vcc, %11, %11, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %3, vcc, %3, %3 " CTRL "\n v_addc_co_u32_dpp %12, vcc, %12, %12, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %4, vcc, %4, %4 " CTRL "\n v_addc_co_u32_dpp %13, vcc, %13, %13, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %5, vcc, %5, %5 " CTRL "\n v_addc_co_u32_dpp %14, vcc, %14, %14, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %6, vcc, %6, %6 " CTRL "\n v_addc_co_u32_dpp %15, vcc, %15, %15, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %7, vcc, %7, %7 " CTRL "\n v_addc_co_u32_dpp %16, vcc, %16, %16, vcc " CTRL "\n" \
                 "v_add_co_u32_dpp %8, vcc, %8, %8 " CTRL "\n v_addc_co_u32_dpp %17, vcc, %17, %17, vcc " CTRL "\n" \
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]), "+v"(lo[8]), \
                   "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]), "+v"(hi[8]) \
                 : : "vcc")
__device__ __forceinline__ void wave_scan9(long long (&v)[9]) {
    unsigned int lo[9], hi[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { lo[j] = static_cast<unsigned int>(v[j]); hi[j] = static_cast<unsigned int>(static_cast<unsigned long long>(v[j]) >> 32); }
    GBRL_SCAN9_STEP("row_shr:1 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9_STEP("row_shr:4 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf");   // lane 15 of rows 0 / 2 -> rows 1 / 3
    GBRL_SCAN9_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf");   // lane 31 -> rows 2, 3
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = static_cast<long long>((static_cast<unsigned long long>(hi[j]) << 32) | lo[j]);
}
#undef GBRL_SCAN9_STEP

// The same scan for nine 32-bit values (sums of at most 4096 fixed-point gradients fit 32 bits: the one-launch growth kernel's LDS histograms):
// one DPP-fused add per value and step.
#define GBRL_SCAN9I_STEP(CTRL) \
    asm volatile("s_nop 1\n" \
                 "v_add_u32_dpp %0, %0, %0 " CTRL "\n v_add_u32_dpp %1, %1, %1 " CTRL "\n v_add_u32_dpp %2, %2, %2 " CTRL "\n" \
                 "v_add_u32_dpp %3, %3, %3 " CTRL "\n v_add_u32_dpp %4, %4, %4 " CTRL "\n v_add_u32_dpp %5, %5, %5 " CTRL "\n" \
                 "v_add_u32_dpp %6, %6, %6 " CTRL "\n v_add_u32_dpp %7, %7, %7 " CTRL "\n v_add_u32_dpp %8, %8, %8 " CTRL "\n" \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]))
__device__ __forceinline__ void wave_scan9(int32_t (&v)[9]) {
    GBRL_SCAN9I_STEP("row_shr:1 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9I_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9I_STEP("row_shr:4 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9I_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
    GBRL_SCAN9I_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf");
    GBRL_SCAN9I_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf");
}
#undef GBRL_SCAN9I_STEP


// One candidate's score from its right-side sums R(d) (exact integers as doubles), the node totals (as doubles) and the side counts:
//   L2: |S_L|^2/n_L + |S_R|^2/n_R (node.cpp:360-373);  Cosine: the square root of it (math_ops.h:538-575).
// Same value, bit for bit, as sum_d ((double)S_d * inv_scale)^2 / n per side: inv_scale is a power of two, so it commutes with every rounding
// (no under- / overflow: |S| < 2^53, inv_scale >= 2^-40) and is applied once, squared, at the end (sqrt: an even power of two);
// (double)(total - right) == (double)total - (double)right because all three are exact.
// (CNT: long long for k_score's row counts; int where they are known to fit -- the conversion to double is one instruction instead of five.)
template <typename GetR, typename CNT>
__device__ __forceinline__ float candidate_score(GetR R, const double *total_f, int D, CNT n_l, CNT n_r, int cosine, double inv_scale) {
    double sr = 0.0, sl_ = 0.0;
#pragma unroll 4
    for (int d = 0; d < D; ++d) {   // (unrolled so that the operand loads of four fields are issued together; the additions keep their order)
        const double vr = R(d);
        const double vl = total_f[d] - vr;
        sr += vr * vr;
        sl_ += vl * vl;
    }
    double x = 0.0;
    if (n_l > 0) x += sl_ / static_cast<double>(n_l);
    if (n_r > 0) x += sr / static_cast<double>(n_r);
    x *= inv_scale * inv_scale;
    return static_cast<float>(cosine ? sqrt(x) : x);
}

}  // namespace
}  // namespace kern
}  // namespace gbrl
