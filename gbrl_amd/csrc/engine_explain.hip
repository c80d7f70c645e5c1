// engine_explain.hip -- Linear TreeSHAP of the ensemble on the device (see shap.hip; host evaluation: explain.cpp).
#include "engine.h"
#include "hooks.h"
#include "explain.h"

#include <cstdlib>
#include <cstring>

namespace gbrl {

namespace {

// depth-first emission of the uniform program: ENTER n; [left subtree; AFTER_LEFT n; right subtree; AFTER_RIGHT n;] EXIT n
void emit(const ShapTree &t, int n, int level, int via, int base, std::vector<kern::ShapOp> &ops) {
    const ShapNode &nd = t.nodes[n];
    ops.push_back({kern::SHAP_ENTER, base + n, level, via});
    if (nd.left >= 0) {
        emit(t, nd.left, level + 1, nd.feature, base, ops);
        ops.push_back({kern::SHAP_AFTER_LEFT, base + n, level, via});
        emit(t, nd.right, level + 1, nd.feature, base, ops);
        ops.push_back({kern::SHAP_AFTER_RIGHT, base + n, level, via});
    }
    if (via >= 0) ops.push_back({kern::SHAP_EXIT, base + n, level, via});
}

}  // namespace

bool Engine::shap_on_device(int tree_idx, const float *obs, const char *cat, int n, const float *norm, const float *base_poly,
                            const float *offset, float *out) {
    const gbrl_hip_metadata &md = model.meta;
    const int D = md.output_dim, depth = md.max_depth, n_num = md.n_num_features, n_cat = md.n_cat_features;
    if (const char *e = hooks::raw(hooks::SHAP_HOST)) { if (e[0] == '1') return false; }   // test hook: host evaluation
    if (kern::shap_block_threads(depth, D) == 0) return false;
    try { ensure_device(); } catch (const NoDeviceError &) { return false; }   // inspection also works on a machine without a GPU
    if (md.n_trees == 0 || n <= 0) return true;
    hipStream_t s = stream_;
    sync_model_to_device();   // categorical conditions -> dictionary ids
    // ---- the program: one tree, or the whole ensemble (cached until the model changes) ----
    const bool whole = tree_idx < 0;
    if (!whole || shap_prog_version_ != model.version) {
        std::vector<kern::ShapOp> ops;
        std::vector<kern::ShapNodeRec> recs;
        std::vector<float> leaf_value;
        const int t0 = whole ? 0 : tree_idx, t1 = whole ? md.n_trees : tree_idx + 1;
        for (int t = t0; t < t1; ++t) {
            const ShapTree st = build_shap_tree(model, t);
            const int node_base = static_cast<int>(recs.size()), value_base = static_cast<int>(leaf_value.size());
            for (size_t i = 0; i < st.nodes.size(); ++i) {
                const ShapNode &nd = st.nodes[i];
                kern::ShapNodeRec r{};
                const bool leaf = nd.left < 0 && nd.right < 0, tied = nd.tied_to >= 0;
                const bool right_child = nd.parent >= 0 && st.nodes[nd.parent].right == static_cast<int>(i);
                r.feature = nd.feature;
                r.flags = (nd.numeric ? 1 : 0) | (tied ? 2 : 0) | (leaf ? 4 : 0) | (right_child ? 8 : 0);
                r.cat_id = (!leaf && !nd.numeric) ? cat_ids_host_[nd.cond] : 0;
                r.pred = leaf ? value_base + nd.pred : 0;
                r.n_unique = nd.n_unique;
                r.n_unique_parent = tied ? st.nodes[nd.tied_to].n_unique : 0;
                r.deg_left = leaf ? 0 : nd.n_unique - st.nodes[nd.left].n_unique;
                r.deg_right = leaf ? 0 : nd.n_unique - st.nodes[nd.right].n_unique;
                r.threshold = nd.threshold;
                r.weight = nd.weight;
                r.weight_parent = tied ? st.nodes[nd.tied_to].weight : 0.0f;
                recs.push_back(r);
            }
            leaf_value.insert(leaf_value.end(), st.leaf_value.begin(), st.leaf_value.end());
            emit(st, 0, 0, -1, node_base, ops);
        }
        leaf_value.push_back(0.0f);
        hip_check(hipMemcpyAsync(d_shap_ops_.ensure(ops.size() * sizeof(kern::ShapOp)), ops.data(), ops.size() * sizeof(kern::ShapOp), hipMemcpyHostToDevice, s), "H2D shap program");
        hip_check(hipMemcpyAsync(d_shap_nodes_.ensure(recs.size() * sizeof(kern::ShapNodeRec)), recs.data(), recs.size() * sizeof(kern::ShapNodeRec), hipMemcpyHostToDevice, s), "H2D shap nodes");
        hip_check(hipMemcpyAsync(d_shap_values_.ensure(leaf_value.size() * 4), leaf_value.data(), leaf_value.size() * 4, hipMemcpyHostToDevice, s), "H2D shap leaf values");
        hip_check(hipStreamSynchronize(s), "sync");   // the host vectors go out of scope
        shap_n_ops_ = static_cast<int>(ops.size());
        shap_prog_version_ = whole ? model.version : ~0ull;
    }
    // ---- inputs ----
    const float *dobs = nullptr;
    if (n_num > 0) {
        float *p = static_cast<float *>(d_pobs_.ensure(sizeof(float) * static_cast<size_t>(n) * n_num));
        hip_check(hipMemcpyAsync(p, obs, sizeof(float) * static_cast<size_t>(n) * n_num, hipMemcpyHostToDevice, s), "H2D obs");
        dobs = p;
    }
    const int32_t *dcat = n_cat > 0 ? encode_categorical_batch(cat, false, n, n_cat) : nullptr;
    const size_t poly = static_cast<size_t>(depth + 1) * depth + depth + static_cast<size_t>(depth) * depth;
    float *dpoly = static_cast<float *>(d_shap_poly_.ensure(poly * 4));
    float *dnorm = dpoly, *dbase = dpoly + static_cast<size_t>(depth + 1) * depth, *doff = dbase + depth;
    hip_check(hipMemcpyAsync(dnorm, norm, sizeof(float) * (depth + 1) * depth, hipMemcpyHostToDevice, s), "H2D norm");
    hip_check(hipMemcpyAsync(dbase, base_poly, sizeof(float) * depth, hipMemcpyHostToDevice, s), "H2D base");
    hip_check(hipMemcpyAsync(doff, offset, sizeof(float) * depth * depth, hipMemcpyHostToDevice, s), "H2D offset");
    const size_t out_bytes = sizeof(float) * static_cast<size_t>(n) * (n_num + n_cat) * D;
    float *dout = static_cast<float *>(d_shap_out_.ensure(out_bytes));
    hip_check(hipMemsetAsync(dout, 0, out_bytes, s), "memset shap");
    kern::shap_values(d_shap_ops_.as<kern::ShapOp>(), shap_n_ops_, d_shap_nodes_.as<kern::ShapNodeRec>(), d_shap_values_.as<float>(), dobs, n_num,
                      dcat, n_cat, n, D, depth, dnorm, dbase, doff, dout, s);
    hip_check(hipGetLastError(), "shap launch");
    hip_check(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, s), "D2H shap");
    hip_check(hipStreamSynchronize(s), "sync");
    return true;
}

}  // namespace gbrl
