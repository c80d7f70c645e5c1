// shap.hip -- Linear TreeSHAP on the device (inspection row f4 of SURVEY.md section 8; semantics: shap.cpp:257-364).
//
// Every sample walks ALL nodes of a tree in the same order, so the walk is a uniform program (ShapOp list, built on the host
// from explain.cpp's ShapTree) that every thread executes without divergence; what differs per thread is data: which side of a
// condition the sample is on and the edge probabilities derived from it.  One thread = one (sample, output); its two coefficient
// stacks C, G [max_depth+1][max_depth] live in LDS with the thread as the fastest index (conflict-free); node records and the
// polynomial vectors are uniform reads (scalar cache).  The arithmetic is the host evaluation's, rounding for rounding
// (explain.cpp Walker: same order of additions, same two fused multiply-adds, IEEE divisions), so both give the same bits.
#include "kernels.h"
#include "kernels_common.h"

#pragma clang fp contract(off)

namespace gbrl {
namespace kern {

namespace {

constexpr int kShapMaxThreads = 256;

__global__ __launch_bounds__(kShapMaxThreads) void k_shap(const ShapOp *__restrict__ ops, int n_ops, const ShapNodeRec *__restrict__ nodes,
                                                       const float *__restrict__ leaf_value, const float *__restrict__ obs, int n_num,
                                                       const int32_t *__restrict__ cat_ids, int n_cat, int n_samples, int D, int md,
                                                       const float *__restrict__ norm, const float *__restrict__ base,
                                                       const float *__restrict__ offset, float *__restrict__ out) {
    extern __shared__ float lds[];
    const int t = threadIdx.x, nt = blockDim.x;
    const int per_block = nt / D;                 // samples per block
    const int s_local = t / D, j = t - s_local * D;
    const long long s = static_cast<long long>(blockIdx.x) * per_block + s_local;
    const bool live = s_local < per_block && s < n_samples;
    // LDS: C [(md+1)*md][nt], G [(md+1)*md][nt], Q [md+1][nt], QA [md+1][nt]
    const int tile = (md + 1) * md;
    float *C = lds + t, *G = lds + tile * nt + t;
    float *Q = lds + 2 * tile * nt + t, *QA = Q + (md + 1) * nt;
    auto at = [nt](float *p, int idx) -> float & { return p[idx * nt]; };
    const float *x = obs ? obs + (live ? s : 0) * n_num : nullptr;
    const int32_t *xc = cat_ids ? cat_ids + (live ? s : 0) * n_cat : nullptr;
    float *phi = out + (live ? s : 0) * static_cast<long long>(n_num + n_cat) * D + j;
    uint32_t act = 0, pass = 0;      // bit k: state of the open node at level k (at most one node per level is open)

    for (int o = 0; o < n_ops; ++o) {
        const ShapOp op = ops[o];
        const ShapNodeRec nd = nodes[op.node];
        const int k = op.level;
        const bool tied = nd.flags & 2, leaf = nd.flags & 4;
        if (op.kind == SHAP_ENTER) {
            bool a = false;
            if (k > 0) { const bool par_pass = (pass >> (k - 1)) & 1u; a = (nd.flags & 8) ? par_pass : !par_pass; }   // flag 8: right child
            float q_anc = 0.0f;
            if (tied) {
                const bool ap = (act >> (k - 1)) & 1u;
                a = a && ap && nd.weight > 0.0f;
                if (ap) q_anc = nd.weight_parent > 0.0f ? __fdiv_rn(1.0f, nd.weight_parent) : 0.0f;
            }
            act = a ? (act | (1u << k)) : (act & ~(1u << k));
            float q = 0.0f;
            if (op.via >= 0) {
                if (a) q = nd.weight > 0.0f ? __fdiv_rn(1.0f, nd.weight) : 0.0f;
                for (int i = 0; i < md; ++i) {
                    float c = at(C, (k - 1) * md + i) * (base[i] + q);
                    if (tied) c = __fdiv_rn(c, base[i] + q_anc);
                    at(C, k * md + i) = c;
                }
            } else {
                for (int i = 0; i < md; ++i) at(C, i) = 1.0f;
            }
            at(Q, k) = q;
            at(QA, k) = q_anc;
            if (leaf) {
                const float pv = leaf_value[nd.pred + j] + 0.0f;
                for (int i = 0; i < md; ++i) at(G, k * md + i) = at(C, k * md + i) * pv;
            } else {
                const bool p = (nd.flags & 1) ? x[nd.feature] > nd.threshold : (xc[nd.feature] == nd.cat_id && nd.cat_id != 0);
                pass = p ? (pass | (1u << k)) : (pass & ~(1u << k));
            }
        } else if (op.kind == SHAP_AFTER_LEFT || op.kind == SHAP_AFTER_RIGHT) {
            const float *off = offset + (op.kind == SHAP_AFTER_LEFT ? nd.deg_left : nd.deg_right) * md;
            for (int i = 0; i < md; ++i) {
                const float g = at(G, (k + 1) * md + i) * (off[i] + 0.0f);
                at(G, (k + 1) * md + i) = g;
                at(G, k * md + i) = op.kind == SHAP_AFTER_LEFT ? g : at(G, k * md + i) + g;
            }
        } else {   // SHAP_EXIT: the edge into this node contributes to the feature it tests (Walker::edge_term)
            if (tied && !((act >> (k - 1)) & 1u)) continue;
            float acc_phi = live ? phi[static_cast<long long>(op.via) * D] : 0.0f;
            for (int pass_no = 0; pass_no < (tied ? 2 : 1); ++pass_no) {
                const int d = pass_no == 0 ? nd.n_unique : nd.n_unique_parent;
                const float qq = pass_no == 0 ? at(Q, k) : at(QA, k);
                const float *nv = norm + d * md;
                const float *off = offset + (pass_no == 0 ? 0 : (nd.n_unique_parent - nd.n_unique) * md);
                const int fused_from = d - d % 4;
                float acc = 0.0f;
                for (int i = 0; i < d; ++i) {
                    const float term = __fdiv_rn(at(G, k * md + i) * off[i], base[i] + qq);
                    if (i < fused_from) { const float pr = term * nv[i]; acc = acc + pr; }
                    else acc = __fmaf_rn(term, nv[i], acc);
                }
                acc = __fdiv_rn(acc, static_cast<float>(d));
                acc_phi = __fmaf_rn(pass_no == 0 ? acc : -acc, qq - 1.0f, acc_phi);
            }
            if (live) phi[static_cast<long long>(op.via) * D] = acc_phi;
        }
    }
}

}  // namespace

// threads per block: the largest of 256 / 128 / 64 whose coefficient stacks fit the LDS and that holds one sample's D outputs; 0 = none
int shap_block_threads(int md, int D) {
    if (md < 1 || md > 31) return 0;
    for (int nt = kShapMaxThreads; nt >= 64; nt >>= 1)
        if (nt >= D && sizeof(float) * nt * (2 * (md + 1) * md + 2 * (md + 1)) <= 156 * 1024) return nt;
    return 0;
}

void shap_values(const ShapOp *ops, int n_ops, const ShapNodeRec *nodes, const float *leaf_value, const float *obs, int n_num,
                 const int32_t *cat_ids, int n_cat, int n_samples, int D, int md, const float *norm, const float *base,
                 const float *offset, float *out, hipStream_t s) {
    if (n_samples <= 0 || n_ops <= 0) return;
    const int nt = shap_block_threads(md, D);
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_shap), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    }
    const int per_block = nt / D;
    const unsigned blocks = static_cast<unsigned>((n_samples + per_block - 1) / per_block);
    const size_t lds = sizeof(float) * nt * (2 * (md + 1) * md + 2 * (md + 1));
    hipLaunchKernelGGL(k_shap, dim3(blocks), dim3(nt), lds, s, ops, n_ops, nodes, leaf_value, obs, n_num, cat_ids, n_cat, n_samples, D, md,
                       norm, base, offset, out);
}

}  // namespace kern
}  // namespace gbrl
