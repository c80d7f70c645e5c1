// oracle/replay_sequence.cpp -- TEST INFRASTRUCTURE ONLY (part of liboracle.so; never linked into the product).
//
// The reference's per-candidate split score and parent score as a float32 OPERATION SEQUENCE, written out so that the bits do not depend on
// this file's compiler (built with -ffp-contract=off; every fused multiply-add is an explicit fmaf): what GCC -O3 makes of node.cpp:187-251
// (splitScoreCosine), :321-376 (splitScoreL2), split_candidate_generator.cpp:262-320 (scoreCosine, scoreL2) and math_ops.h:432-449,
// 476-485, 538-575 for an AVX2 + FMA target (the reference's release flags, oracle/Makefile):
//   * the `omp simd` reductions without a reduction clause stay IN ORDER; their vectorised part (columns below n_cols & ~3) rounds the
//     product before the add, the scalar remainder (the last n_cols % 4 columns) is contracted to fmaf;
//   * Cosine: den = fmaf(norm_right, n_right, norm_left * n_left); L2: fmaf(n_left, norm_left, n_right * norm_right).
// Pinned bit for bit against the reference's own functions (oracle/_ref/ref_score_probe.so) by tests/test_oracle.py; the product's near-tie
// replay (gbrl_amd/csrc/neartie.hip) is tested against both.
#include <cmath>
#include <cstring>
#include <vector>

static float dot_rows(const int *idx, int n, const float *g, const float *vec, int D) {
    const int D4 = D & ~3;
    float s = 0.0f;
    for (int r = 0; r < n; ++r) {
        const float *row = g + (size_t)idx[r] * D;
        for (int c = 0; c < D4; ++c) { float p = row[c] * vec[c]; s = s + p; }
        for (int c = D4; c < D; ++c) s = fmaf(row[c], vec[c], s);
    }
    return s;
}
static float sqnorm(const float *v, int D) {
    const int D4 = D & ~3;
    float s = 0.0f;
    for (int c = 0; c < D4; ++c) { float p = v[c] * v[c]; s = s + p; }
    for (int c = D4; c < D; ++c) s = fmaf(v[c], v[c], s);
    return s;
}
extern "C" float oracle_replay_split_cosine(const float *obs, const float *g, const int *rows, int n, int F, int D, int f, float v, int min_data) {
    std::vector<float> lm(D, 0.f), rm(D, 0.f);
    std::vector<int> li, ri;
    for (int k = 0; k < n; ++k) {
        int r = rows[k];
        if (obs[(size_t)r * F + f] > v) { for (int d = 0; d < D; ++d) rm[d] += g[(size_t)r * D + d]; ri.push_back(r); }
        else { for (int d = 0; d < D; ++d) lm[d] += g[(size_t)r * D + d]; li.push_back(r); }
    }
    int nl = li.size(), nr = ri.size();
    if (nl < min_data || nr < min_data) return -INFINITY;
    float nlf = nl, nrf = nr;
    float lrec = nl > 0 ? 1.0f / nlf : 0.0f, rrec = nr > 0 ? 1.0f / nrf : 0.0f;
    for (int d = 0; d < D; ++d) { lm[d] *= lrec; rm[d] *= rrec; }
    float tnum = 0.f, fnum = 0.f;
    if (nr > 0) tnum = dot_rows(ri.data(), nr, g, rm.data(), D);
    if (nl > 0) fnum = dot_rows(li.data(), nl, g, lm.data(), D);
    float tn = sqnorm(rm.data(), D), fn = sqnorm(lm.data(), D);
    float fden = fn * nlf;
    float den = fmaf(tn, nrf, fden);
    float num = tnum + fnum;
    if (den == 0.0f) return 0.0f;
    return num / sqrtf(den);
}
extern "C" float oracle_replay_parent_cosine(const float *g, const int *rows, int n, int D) {
    std::vector<float> m(D, 0.f);
    float nf = n, rec = 1.0f / nf;
    for (int k = 0; k < n; ++k) for (int d = 0; d < D; ++d) m[d] += g[(size_t)rows[k] * D + d];
    for (int d = 0; d < D; ++d) m[d] *= rec;
    if (n == 0) return 0.0f;
    float dot = dot_rows(rows, n, g, m.data(), D);
    float den = sqnorm(m.data(), D) * nf;
    if (den == 0.0f) return 0.0f;
    return (float)((double)dot / sqrt((double)den));
}
extern "C" float oracle_replay_split_l2(const float *obs, const float *g, const int *rows, int n, int F, int D, int f, float v, int min_data) {
    std::vector<float> lm(D, 0.f), rm(D, 0.f);
    int nl = 0, nr = 0;
    for (int k = 0; k < n; ++k) {
        int r = rows[k];
        if (obs[(size_t)r * F + f] > v) { for (int d = 0; d < D; ++d) rm[d] += g[(size_t)r * D + d]; ++nr; }
        else { for (int d = 0; d < D; ++d) lm[d] += g[(size_t)r * D + d]; ++nl; }
    }
    if (nl < min_data || nr < min_data) return -INFINITY;
    float nlf = nl, nrf = nr;
    float lrec = nl > 0 ? 1.0f / nlf : 0.0f, rrec = nr > 0 ? 1.0f / nrf : 0.0f;
    for (int d = 0; d < D; ++d) { lm[d] *= lrec; rm[d] *= rrec; }
    float ln = sqnorm(lm.data(), D), rn = sqnorm(rm.data(), D);
    const float rp = nrf * rn;
    return fmaf(nlf, ln, rp);
}
extern "C" float oracle_replay_parent_l2(const float *g, const int *rows, int n, int D) {
    std::vector<float> m(D, 0.f);
    float nf = n, rec = 1.0f / nf;
    for (int k = 0; k < n; ++k) for (int d = 0; d < D; ++d) m[d] += g[(size_t)rows[k] * D + d];
    for (int d = 0; d < D; ++d) m[d] *= rec;
    return sqnorm(m.data(), D) * nf;
}

// ---- the same sequence on explicit side lists (rows in node order): what oracle.cpp's split_score / parent scores call, so that the
// restatement's bits do not depend on how oracle.cpp's own compiler contracts "a * b + c * d" (the two sides are NOT interchangeable:
// mirror-image candidates -- "== a" against "== b" on a two-token column -- tie exactly, and the last bit of the asymmetric expression decides)
extern "C" float oracle_seq_split(const float *g, const int *ridx, int nr, const int *lidx, int nl, int D, int cosine) {
    std::vector<float> lm(D, 0.f), rm(D, 0.f);
    // (node.cpp:196-222 / 336-352: ONE pass over the node's rows adds each to its side; per side that is the side's rows in node order)
    for (int k = 0; k < nr; ++k) for (int d = 0; d < D; ++d) rm[d] += g[(size_t)ridx[k] * D + d];
    for (int k = 0; k < nl; ++k) for (int d = 0; d < D; ++d) lm[d] += g[(size_t)lidx[k] * D + d];
    float nlf = nl, nrf = nr;
    float lrec = nl > 0 ? 1.0f / nlf : 0.0f, rrec = nr > 0 ? 1.0f / nrf : 0.0f;
    for (int d = 0; d < D; ++d) { lm[d] *= lrec; rm[d] *= rrec; }
    if (!cosine) {
        float ln = sqnorm(lm.data(), D), rn = sqnorm(rm.data(), D);
        const float rp = nrf * rn;
        return fmaf(nlf, ln, rp);
    }
    float tnum = 0.f, fnum = 0.f;
    if (nr > 0) tnum = dot_rows(ridx, nr, g, rm.data(), D);
    if (nl > 0) fnum = dot_rows(lidx, nl, g, lm.data(), D);
    float tn = sqnorm(rm.data(), D), fn = sqnorm(lm.data(), D);
    float fden = fn * nlf;
    float den = fmaf(tn, nrf, fden);
    float num = tnum + fnum;
    if (den == 0.0f) return 0.0f;
    return num / sqrtf(den);
}
