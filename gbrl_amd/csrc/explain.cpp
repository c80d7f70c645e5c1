// explain.cpp -- SHAP values, C-header export and text dumps from the host copy of the ensemble (see explain.h).
//
// The outputs are defined by the reference and are reproduced INCLUDING its quirks, because
// the unchanged Python package and its users compare against them:
//   * SHAP explains the raw leaf values (no learning rate, no sign flip, no bias) and adds a categorical condition's
//     contribution to row `feature_indices` of the output, i.e. WITHOUT the n_num_features offset (shap.cpp:286, 324);
//   * "unique features on a path" counts raw feature indices, numeric and categorical alike (shap.cpp:108, utils.cpp:90-110);
//   * a node whose parent's feature re-occurs higher up is tied to its PARENT (shap.cpp:124-141), not to that ancestor;
//   * the export writes the first sum(depths) entries of the max_depth-strided condition arrays (types.cpp:564-590), which
//     is the intended packing only when every tree reached max_depth, glues "iteration" to "alloc_data_size" in the comment
//     block, leaves N_INPUTS un-prefixed and spells the grow policy "Oblivous" (types.cpp:112-116);
//   * print_tree writes all 128 bytes of a categorical value, NULs included (node.cpp:531-534).
#include "explain.h"

// the SHAP sums below spell out every rounding (see Walker::edge_term): keep the compiler from fusing anything else
#pragma clang fp contract(off)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

namespace gbrl {
namespace {

// ------------------------------------------------------------------------------------------------------------------
// SHAP: one explicit binary tree per ensemble member, nodes numbered in depth-first (left before right) order
// ------------------------------------------------------------------------------------------------------------------
using XNode = ShapNode;
using XTree = ShapTree;

int distinct_count(const int32_t *v, int n) {   // utils.cpp:90-110: 1 for an empty path
    int c = 1;
    for (int i = 1; i < n; ++i) {
        bool seen = false;
        for (int j = 0; j < i && !seen; ++j) seen = v[j] == v[i];
        c += seen ? 0 : 1;
    }
    return c;
}

struct TreeBuilder {
    const Model &m;
    const int tree, md, D;
    const int first_leaf, end_leaf;
    int next_leaf;
    XTree out;

    TreeBuilder(const Model &model, int t)
        : m(model), tree(t), md(model.meta.max_depth), D(model.meta.output_dim), first_leaf(model.tree_indices[t]),
          end_leaf(t == model.meta.n_trees - 1 ? model.meta.n_leaves : model.tree_indices[t + 1]), next_leaf(first_leaf) {}

    // conditions live per tree (oblivious) or per leaf (greedy); a node reads them from the left-most leaf below it, which is
    // the leaf the running counter points at when the node is entered (shap.cpp:77, 90-99)
    int cond_row() const {
        if (next_leaf >= end_leaf) throw std::runtime_error("tree_shap: the leaves of the tree are not stored in depth-first order");
        return m.oblivious() ? tree : next_leaf;
    }

    int add(int parent, int depth) {
        const int n = static_cast<int>(out.nodes.size());
        out.nodes.emplace_back();
        const int row = cond_row();
        {
            XNode &nd = out.nodes[n];
            nd.parent = parent;
            if (depth > 0) nd.weight = m.edge_weights[static_cast<size_t>(next_leaf) * md + depth - 1];
        }
        const int tree_depth = m.depths[row];
        if (depth < tree_depth) {
            const size_t c = static_cast<size_t>(row) * md + depth;
            XNode &nd = out.nodes[n];
            nd.feature = m.feature_indices[c];
            nd.numeric = m.is_numerics[c] != 0;
            if (nd.numeric) nd.threshold = m.feature_values[c];
            else nd.cond = static_cast<int>(c);
        } else {
            const int leaf = next_leaf++;
            const int uniq = distinct_count(m.feature_indices.data() + static_cast<size_t>(row) * md, tree_depth);
            for (int a = n; a >= 0; a = out.nodes[a].parent) out.nodes[a].n_unique = std::max(out.nodes[a].n_unique, uniq);
            float cover = 1.0f;
            for (int d = 0; d < tree_depth; ++d) cover *= m.edge_weights[static_cast<size_t>(leaf) * md + d];
            out.nodes[n].pred = static_cast<int>(out.leaf_value.size());
            for (int d = 0; d < D; ++d) out.leaf_value.push_back(m.values[static_cast<size_t>(leaf) * D + d] * cover);
        }
        // does the parent's feature occur again among the parent's own ancestors?
        if (parent >= 0) {
            const int f = out.nodes[parent].feature;
            bool again = false;
            for (int a = out.nodes[parent].parent; a >= 0 && !again; a = out.nodes[a].parent) again = out.nodes[a].feature == f;
            if (again) {
                out.nodes[n].tied_to = parent;
                out.nodes[n].weight *= out.nodes[parent].weight;
            }
        }
        if (depth < tree_depth) {
            const int l = add(n, depth + 1);
            out.nodes[n].left = l;
            const int r = add(n, depth + 1);
            out.nodes[n].right = r;
        }
        return n;
    }
};

XTree build_tree(const Model &m, int t) { return build_shap_tree(m, t); }

// per-sample evaluation state: two stacks of [max_depth][D] coefficient tables, one row per tree level
struct Walker {
    const Model &m;
    const float *norm, *base, *offset;
    const int md, D, tile;
    std::vector<float> C, G;
    std::vector<uint8_t> active;
    const XTree *T = nullptr;
    const float *x = nullptr;
    const char *xc = nullptr;
    float *phi = nullptr;   // [n_features][D] of the current sample

    Walker(const Model &model, const float *norm_values, const float *base_poly, const float *offset_poly)
        : m(model), norm(norm_values), base(base_poly), offset(offset_poly), md(model.meta.max_depth), D(model.meta.output_dim),
          tile(md * D), C(static_cast<size_t>(md + 1) * tile), G(static_cast<size_t>(md + 1) * tile) {}

    void run(const XTree &tree, const float *obs_row, const char *cat_row, float *out_row) {
        T = &tree; x = obs_row; xc = cat_row; phi = out_row;
        active.assign(tree.nodes.size(), 0);
        std::fill(C.begin(), C.end(), 0.0f);
        std::fill(G.begin(), G.end(), 0.0f);
        std::fill(C.begin(), C.begin() + tile, 1.0f);
        visit(0, 0, -1);
    }

    // dst[j] +/-= (sum_i G[i][j] * off[i] / (base[i] + q) * nv[i]) / d * (q - 1)   for every output j   (shap.cpp:340-364).
    // In float32 this sum cancels badly from max_depth 5 on (the normalisation rows reach 1e3..1e4: the reference's own values are
    // ~10% of the array's scale away from a float64 evaluation at depth 6), so its value depends on how each product is rounded.
    // The terms are added in index order; the reference's -O3 FMA build (src/cpp/CMakeLists.txt:89) vectorises the
    // loop four terms at a time with separate multiply and add and finishes the d mod 4 trailing terms with fused multiply-adds,
    // and fuses the final scaling.  The same roundings are made here so that the values agree with it.
    void edge_term(float *dst, const float *g, const float *off, float q, const float *nv, int d, bool subtract) const {
        const int fused_from = d - d % 4;
        for (int j = 0; j < D; ++j) {
            float acc = 0.0f;
            for (int i = 0; i < d; ++i) {
                const float t = g[i * D + j] * off[i] / (base[i] + q);
                if (i < fused_from) { const float p = t * nv[i]; acc += p; }
                else acc = std::fmaf(t, nv[i], acc);
            }
            acc /= static_cast<float>(d);
            dst[j] = std::fmaf(subtract ? -acc : acc, q - 1.0f, dst[j]);
        }
    }

    void visit(int n, int level, int via_feature) {
        const XNode &nd = T->nodes[n];
        const int tied = nd.tied_to;
        float q_anc = 0.0f;
        if (tied >= 0) {
            active[n] = active[n] && active[tied] && nd.weight > 0.0f;
            if (active[tied]) q_anc = T->nodes[tied].weight > 0.0f ? 1.0f / T->nodes[tied].weight : 0.0f;
        }
        float *g = G.data() + static_cast<size_t>(level) * tile;
        float *c = C.data() + static_cast<size_t>(level) * tile;
        float q = 0.0f;
        if (via_feature >= 0) {
            if (active[n]) q = nd.weight > 0.0f ? 1.0f / nd.weight : 0.0f;
            const float *c_up = c - tile;
            for (int i = 0; i < md; ++i)
                for (int j = 0; j < D; ++j) c[i * D + j] = c_up[i * D + j] * (base[i] + q);
            if (tied >= 0)
                for (int i = 0; i < md; ++i)
                    for (int j = 0; j < D; ++j) c[i * D + j] /= (base[i] + q_anc);
        }
        if (nd.left < 0 && nd.right < 0) {
            const float *pv = T->leaf_value.data() + nd.pred;
            for (int i = 0; i < md; ++i)
                for (int j = 0; j < D; ++j) g[i * D + j] = c[i * D + j] * (pv[j] + 0.0f);
        } else {
            const bool pass = nd.numeric ? x[nd.feature] > nd.threshold
                                         : std::strcmp(xc + static_cast<size_t>(nd.feature) * kCat, m.categorical_values.data() + static_cast<size_t>(nd.cond) * kCat) == 0;
            active[nd.right] = pass;
            active[nd.left] = !pass;
            float *g_dn = g + tile;
            visit(nd.left, level + 1, nd.feature);
            const float *off = offset + static_cast<size_t>(nd.n_unique - T->nodes[nd.left].n_unique) * md;
            for (int i = 0; i < md; ++i)
                for (int j = 0; j < D; ++j) { g_dn[i * D + j] *= (off[i] + 0.0f); g[i * D + j] = g_dn[i * D + j]; }
            visit(nd.right, level + 1, nd.feature);
            off = offset + static_cast<size_t>(nd.n_unique - T->nodes[nd.right].n_unique) * md;
            for (int i = 0; i < md; ++i)
                for (int j = 0; j < D; ++j) { g_dn[i * D + j] *= (off[i] + 0.0f); g[i * D + j] += g_dn[i * D + j]; }
        }
        if (via_feature < 0) return;
        if (tied >= 0 && !active[tied]) return;
        float *dst = phi + static_cast<size_t>(via_feature) * D;
        edge_term(dst, g, offset, q, norm + static_cast<size_t>(nd.n_unique) * md, nd.n_unique, false);
        if (tied >= 0) {
            const int du = T->nodes[tied].n_unique;
            edge_term(dst, g, offset + static_cast<size_t>(du - nd.n_unique) * md, q_anc, norm + static_cast<size_t>(du) * md, du, true);
        }
    }
};

void check_tree_index(const Model &m, int t) {   // valid_tree_idx utils.h:101-108 (which also lets t == n_trees through; not reproduced)
    if (t < 0 || t >= m.meta.n_trees) throw std::runtime_error("Invalid tree index");
}

void shap_over(const Model &m, const std::vector<XTree> &trees, const float *obs, const char *cat_obs, int n_samples, const float *norm,
               const float *base, const float *offset, float *out) {
    const int n_num = m.meta.n_num_features, n_cat = m.meta.n_cat_features, D = m.meta.output_dim;
    const size_t row = static_cast<size_t>(n_num + n_cat) * D;
    size_t nodes = 0;
    for (const XTree &t : trees) nodes += t.nodes.size();
    const size_t work = nodes * static_cast<size_t>(n_samples);
    unsigned n_thr = std::max(1u, std::thread::hardware_concurrency());
    n_thr = static_cast<unsigned>(std::min<size_t>({n_thr, static_cast<size_t>(std::max(1, n_samples)), work / 20000 + 1}));
    auto body = [&](int lo, int hi) {
        Walker w(m, norm, base, offset);
        for (int s = lo; s < hi; ++s)
            for (const XTree &t : trees)
                w.run(t, obs ? obs + static_cast<size_t>(s) * n_num : nullptr, cat_obs ? cat_obs + static_cast<size_t>(s) * n_cat * kCat : nullptr,
                      out + static_cast<size_t>(s) * row);
    };
    if (n_thr <= 1) { body(0, n_samples); return; }
    std::vector<std::thread> pool;
    const int per = (n_samples + static_cast<int>(n_thr) - 1) / static_cast<int>(n_thr);
    for (unsigned t = 0; t < n_thr; ++t) {
        const int lo = static_cast<int>(t) * per, hi = std::min(n_samples, lo + per);
        if (lo < hi) pool.emplace_back(body, lo, hi);
    }
    for (auto &th : pool) th.join();
}

void check_shap_inputs(const Model &m, const float *obs, const char *cat_obs, const float *norm, const float *base, const float *offset) {
    if (!norm || !base || !offset) throw std::runtime_error("tree_shap: norm_values, base_poly and offset are required");
    if (m.meta.n_num_features > 0 && !obs) throw std::runtime_error("tree_shap: the model has numerical features but obs is None");
    if (m.meta.n_cat_features > 0 && !cat_obs) throw std::runtime_error("tree_shap: the model has categorical features but categorical_obs is None");
}

// ------------------------------------------------------------------------------------------------------------------
// text
// ------------------------------------------------------------------------------------------------------------------
struct Text {
    std::string s;
    Text &operator<<(const std::string &v) { s += v; return *this; }
    Text &operator<<(const char *v) { s += v; return *this; }
    Text &operator<<(char v) { s.push_back(v); return *this; }
    Text &operator<<(int v) { s += std::to_string(v); return *this; }
    Text &operator<<(long v) { s += std::to_string(v); return *this; }
    Text &operator<<(size_t v) { s += std::to_string(v); return *this; }
    Text &operator<<(float v) {   // std::ostream default: %g with 6 significant digits
        char b[48];
        std::snprintf(b, sizeof b, "%g", static_cast<double>(v));
        s += b;
        return *this;
    }
};

const char *score_name(int v) { return v == GBRL_HIP_SCORE_L2 ? "L2" : "Cosine"; }
const char *generator_name(int v) { return v == GBRL_HIP_GEN_UNIFORM ? "Uniform" : "Quantile"; }
const char *policy_name(int v) { return v == GBRL_HIP_GROW_OBLIVIOUS ? "Oblivous" : "Greedy"; }   // sic, types.cpp:115

long fixed_point(float v, int frac_bits, float lo, float hi) {   // float_to_int16 / float_to_int32 utils.h:203-240
    float scaled = v * static_cast<float>(1 << frac_bits);
    scaled = std::max(scaled, lo);
    scaled = std::min(scaled, hi);
    const float r = std::round(scaled);
    // the reference converts a float that can be 2147483648.0f after the clamp (static_cast<float>(INT32_MAX) rounds up): out of range
    // for int32; its build prints INT32_MAX there (seen for the +inf thresholds of categorical conditions)
    if (r >= 2147483648.0f) return static_cast<long>(INT32_MAX);
    return static_cast<long>(r);
}

enum class Fmt { Float, Fxp8, Fxp16 };

void put_number(Text &t, Fmt f, float v) {
    if (f == Fmt::Float) t << v;
    else if (f == Fmt::Fxp8) t << fixed_point(v, 8, -32768.0f, 32767.0f);
    else t << fixed_point(v, 16, static_cast<float>(INT32_MIN), static_cast<float>(INT32_MAX));
}

}  // namespace

void check_shap_arguments(const Model &m, int tree_idx, const float *obs, const char *cat_obs, const float *norm_values,
                          const float *base_poly, const float *offset) {
    check_tree_index(m, tree_idx);
    check_shap_inputs(m, obs, cat_obs, norm_values, base_poly, offset);
}

ShapTree build_shap_tree(const Model &m, int t) {
    if (t < 0 || t >= m.meta.n_trees) throw std::runtime_error("Invalid tree index");
    TreeBuilder b(m, t);
    b.add(-1, 0);
    return std::move(b.out);
}

// ------------------------------------------------------------------------------------------------------------------
void tree_shap(const Model &m, int tree_idx, const float *obs, const char *cat_obs, int n_samples, const float *norm_values,
               const float *base_poly, const float *offset, float *out) {
    check_tree_index(m, tree_idx);
    check_shap_inputs(m, obs, cat_obs, norm_values, base_poly, offset);
    std::vector<XTree> one;
    one.push_back(build_tree(m, tree_idx));
    shap_over(m, one, obs, cat_obs, n_samples, norm_values, base_poly, offset, out);
}

void ensemble_shap(const Model &m, const float *obs, const char *cat_obs, int n_samples, const float *norm_values,
                   const float *base_poly, const float *offset, float *out) {
    if (m.meta.n_trees == 0) return;   // the reference returns zeros for an empty ensemble (gbrl.cpp:1306, 1330)
    check_shap_inputs(m, obs, cat_obs, norm_values, base_poly, offset);
    std::vector<XTree> all;
    all.reserve(m.meta.n_trees);
    for (int t = 0; t < m.meta.n_trees; ++t) all.push_back(build_tree(m, t));
    shap_over(m, all, obs, cat_obs, n_samples, norm_values, base_poly, offset, out);
}

size_t reference_alloc_bytes(const Model &m, bool compact) {
    const gbrl_hip_metadata &md = m.meta;
    const size_t trees = compact ? md.n_trees : md.max_trees, leaves = compact ? md.n_leaves : md.max_leaves;
    const size_t in = md.input_dim, D = md.output_dim, depth = md.max_depth;
    const size_t rows = m.oblivious() ? trees : leaves;   // split_sizes
    size_t b = 0;
    b += 4 * D + 4 * in;                          // bias, feature_weights
    b += 4 * trees + 4 * rows + 4 * leaves * D;   // tree_indices, depths, values
    b += (4 + 4) * rows * depth;                  // feature_indices, feature_values
    b += 4 * leaves * depth;                      // edge_weights
    b += 3 * 4 * in + in;                         // the three int mappings + mapping_numerics
    b += rows * depth + leaves * depth;           // is_numerics, inequality_directions
    b += rows * depth * kCat;                     // categorical_values
    return b;
}

bool export_header(const Model &m, const std::string &model_name, const std::string &export_format, const std::string &export_type,
                   const std::string &prefix, std::string &text) {
    if (!m.oblivious()) throw std::runtime_error("Export is supported only for Oblivious trees.");   // gbrl.cpp:1113-1118
    Fmt fmt;
    if (export_format == "float") fmt = Fmt::Float;
    else if (export_format == "fxp8") fmt = Fmt::Fxp8;
    else if (export_format == "fxp16") fmt = Fmt::Fxp16;
    else throw std::runtime_error("Invalid exportFormat! Options are: float/fxp8/fxp16");             // types.cpp:64-69
    bool compact;
    if (export_type == "compact") compact = true;
    else if (export_type == "full") compact = false;
    else throw std::runtime_error("Invalid exportType Options are: full/compact");                    // types.cpp:70-74
    const gbrl_hip_metadata &md = m.meta;
    text.clear();
    if (compact && md.max_depth > 6) {
        std::fprintf(stderr, "Cannot only compact export with max depth <= 6 and oblivious trees\n");
        return false;
    }
    for (const gbrl_hip_optimizer &o : m.opts)
        if (o.algo != GBRL_HIP_ALGO_SGD) throw std::runtime_error("Error. Can only export SGD optimizers");

    const char *ctype = fmt == Fmt::Float ? "float" : fmt == Fmt::Fxp8 ? "int16" : "int32";
    const int T = md.n_trees, L = md.n_leaves, D = md.output_dim;
    int n_conditions = 0;
    for (int t = 0; t < T; ++t) n_conditions += m.depths[t];
    const bool vector_out = D > 1;

    Text h;
    h << "#ifndef GBRL_MODEL_H\n#define GBRL_MODEL_H\n\n/*\n";
    if (!model_name.empty()) h << "###########################\nmodel_name: " << model_name << "\n";
    h << "###########################\n";
    // comment block: "name: value" pairs, a line break before max_depth and before split_score_func
    struct Pair { const char *name; int value; };
    const Pair sizes[] = {{"n_leaves", L}, {"n_trees", T}, {"max_trees", md.max_trees}, {"max_leaves", md.max_leaves},
                          {"max_trees_batch", md.max_trees_batch}, {"max_leaves_batch", md.max_leaves_batch}, {"input_dim", md.input_dim},
                          {"output_dim", D}, {"policy_dim", md.policy_dim}};
    for (const Pair &p : sizes) h << p.name << ": " << p.value << ", ";
    h << "\n";
    const Pair growth[] = {{"max_depth", md.max_depth}, {"min_data_in_leaf", md.min_data_in_leaf}, {"n_bins", md.n_bins}, {"par_th", md.par_th}};
    for (const Pair &p : growth) h << p.name << ": " << p.value << ", ";
    h << "cv_beta: " << md.cv_beta << ", verbose: " << md.verbose << ", batch_size: " << md.batch_size << ", use_cv: " << static_cast<int>(md.use_cv);
    h << "\nsplit_score_func: " << score_name(md.split_score_func) << ", generator_type: " << generator_name(md.generator_type)
      << ", grow_policy: " << policy_name(md.grow_policy) << ", n_num_features: " << md.n_num_features << ", n_cat_features: "
      << md.n_cat_features << ", iteration: " << md.iteration << "alloc_data_size: " << reference_alloc_bytes(m, false) << "\n*/\n";

    const Pair counts[] = {{"N_TREES", T}, {"N_LEAVES", L}, {"BINARY_FEATURES", n_conditions}};
    for (const Pair &p : counts) h << "#define " << prefix << p.name << " " << p.value << "\n";
    h << "#define N_INPUTS " << md.input_dim << "\n";
    h << "#define " << prefix << "N_OUTPUTS " << D << "\n#define " << prefix << "N_FEATURES " << md.n_num_features << "\n\n";

    if (vector_out) {
        h << "static inline void gbrl_predict(" << ctype << " *results, const " << ctype << " *features){\n\n";
    } else {
        h << "static inline " << ctype << " gbrl_predict(const " << ctype << " *features){\n\n\t" << ctype << " result = "
          << (fmt == Fmt::Float ? "0.0f" : "0") << ";\n";
    }
    h << "\tunsigned int tree_idx, idx, leaf_ptr, cond_ptr" << (vector_out ? ", j" : "") << (compact ? "" : ", depth, current_depth") << ";\n";
    h << "\t/* Model data */\n";

    auto list = [&](int n, auto &&item) {   // "a, b, c"
        for (int i = 0; i < n; ++i) { item(i); if (i + 1 < n) h << ", "; }
    };
    if (!compact) {
        h << "\tconst unsigned int depths[" << prefix << "N_TREES] = {";
        list(T, [&](int i) { h << static_cast<int>(m.depths[i]); });
        h << "};\n";
    }
    if (vector_out) {
        h << "\tconst " << ctype << " bias[" << prefix << "N_OUTPUTS] = {";
        list(D, [&](int i) { put_number(h, fmt, m.bias[i]); });
    } else {
        h << "\tconst " << ctype << " bias = ";
        put_number(h, fmt, m.bias[0]);
    }
    h << ";\n";   // (the vector form is left without its closing brace by the reference, types.cpp:516-548)
    h << "\tconst " << (md.input_dim < 255 ? "uint8" : "uint16") << " feature_indices[" << prefix << "BINARY_FEATURES] = {";
    list(n_conditions, [&](int i) { h << static_cast<int>(m.feature_indices[i]); });
    h << "};\n\tconst " << ctype << " feature_values[" << prefix << "BINARY_FEATURES] = {";
    list(n_conditions, [&](int i) { put_number(h, fmt, m.feature_values[i]); });
    h << "};\n\tconst " << ctype << " leaf_values[" << prefix << "N_LEAVES*" << prefix << "N_OUTPUTS]  = {";
    // every optimizer contributes its own output range, scaled by -lr (constant schedules only: model.cpp add_optimizer)
    for (int leaf = 0; leaf < L; ++leaf)
        for (const gbrl_hip_optimizer &o : m.opts)
            for (int j = o.start_idx; j < o.stop_idx; ++j) {
                put_number(h, fmt, -m.values[static_cast<size_t>(leaf) * D + j] * o.init_lr);
                if (leaf < L - 1 || j < D - 1) h << ", ";
            }
    h << "};\n\tleaf_ptr = 0;\n\tcond_ptr = 0;\n\tunsigned char pass;\n\tfor (tree_idx = 0; tree_idx < " << prefix << "N_TREES; ++tree_idx)\n\t{\n";
    if (compact) {
        h << "\t\tidx = 0;\n";
        for (int d = 0; d < md.max_depth; ++d) {
            h << "\t\tpass = (unsigned char)(features[feature_indices[cond_ptr + " << d << "]] > feature_values[cond_ptr + " << d << "]);\n";
            h << "\t\tidx |= (pass <<  (" << md.max_depth << " - 1 - " << d << "));\n";
        }
    } else {
        h << "\t\tcurrent_depth = depths[tree_idx];\n\t\tidx = 0;\n\t\tfor (depth = 0; depth < current_depth; ++depth){\n"
             "\t\t\tpass = (unsigned char)(features[feature_indices[cond_ptr + depth]] > feature_values[cond_ptr + depth]);\n"
             "\t\t\tidx |= (pass <<  (current_depth - 1 - depth));\n\t\t}\n";
    }
    if (vector_out)
        h << "\t\tfor (j = 0 ; j < " << prefix << "N_OUTPUTS; j++)\n\t\t\tresults[j] += leaf_values[(leaf_ptr + idx)*" << prefix << "N_OUTPUTS + j];\n";
    else
        h << "\t\tresult += leaf_values[leaf_ptr + idx];\n";
    if (compact) h << "\t\tleaf_ptr += " << (1 << md.max_depth) << ";\n\t\tcond_ptr += " << md.max_depth << ";\n";
    else h << "\t\tleaf_ptr += (1 << current_depth);\n\t\tcond_ptr += current_depth;\n";
    h << "\t}\n";
    if (vector_out) h << "\tfor (j = 0 ; j < " << prefix << "N_OUTPUTS; j++)\n\t\tresults[j] += bias[j];\n";
    else h << "\tresult += bias;\n\treturn result;\n";
    h << "}\n#endif\n";
    text = std::move(h.s);
    return true;
}

std::string tree_text(const Model &m, int tree_idx) {
    const gbrl_hip_metadata &md = m.meta;
    Text o;
    if (tree_idx == -1) {
        tree_idx = md.n_trees - 1;
        o << "No tree index provided. Printing last tree in the ensemble containing " << md.n_trees << " trees\n";
    }
    if (tree_idx < 0 || tree_idx >= md.n_trees) throw std::runtime_error("Invalid tree index");
    const int first = m.tree_indices[tree_idx];
    const int end = tree_idx == md.n_trees - 1 ? md.n_leaves : m.tree_indices[tree_idx + 1];
    const int depth_cap = md.max_depth;
    o << policy_name(md.grow_policy) << " DecisionTree idx: " << tree_idx << " output_dim: " << md.output_dim << " n_bins: " << md.n_bins
      << " min_data_in_leaf: " << md.min_data_in_leaf << " par_th: " << md.par_th << " max_depth: " << depth_cap << "\n";
    o << " input_dim: " << md.input_dim << " with " << md.n_num_features << " numerical features and " << md.n_cat_features
      << " categorical features\n";
    o << "Leaf Nodes: " << (end - first) << "\n";
    for (int leaf = first; leaf < end; ++leaf) {
        const int row = m.oblivious() ? tree_idx : leaf;
        const int depth = m.depths[row];
        const size_t cond = static_cast<size_t>(row) * depth_cap, path = static_cast<size_t>(leaf) * depth_cap;
        auto bracket = [&](const char *title, auto &&item, const char *close) {
            o << title;
            for (int i = 0; i < depth; ++i) { item(i); if (i + 1 < depth) o << ", "; }
            o << close;
        };
        o << "Leaf idx: " << (leaf - first) << " tree_idx: " << tree_idx << " output_dim: " << md.output_dim << " depth: " << depth << " value: [";
        for (int d = 0; d < md.output_dim; ++d) { o << m.values[static_cast<size_t>(leaf) * md.output_dim + d]; if (d + 1 < md.output_dim) o << ", "; }
        o << "] ";
        bracket(" feature_idxs: [", [&](int i) { o << (m.feature_indices[cond + i] + (m.is_numerics[cond + i] ? 0 : md.n_num_features)); }, "] ");
        bracket(" inequality_directions: [", [&](int i) { o << static_cast<int>(m.inequality_directions[path + i]); }, "] ");
        bracket(" feature_values: [", [&](int i) {
            if (m.is_numerics[cond + i]) o << m.feature_values[cond + i];
            else o.s.append(m.categorical_values.data() + (cond + i) * kCat, kCat);
        }, "]\n");
        bracket(" edge_weights: [", [&](int i) { o << m.edge_weights[path + i]; }, "]\n");
    }
    o << "******************\n";
    return std::move(o.s);
}

std::string metadata_text(const Model &m, const char *device_name) {
    const gbrl_hip_metadata &md = m.meta;
    Text o;
    o << "######## " << m.learner_name << " model ########\n";
    o << "input dim: " << md.input_dim << " output dim: " << md.output_dim << " policy dim: " << md.policy_dim << " max depth: " << md.max_depth
      << " min data in leaf: " << md.min_data_in_leaf << "\n";
    o << "generator type: " << generator_name(md.generator_type) << " n bins: " << md.n_bins << " cv beta: " << md.cv_beta
      << " split score func: " << score_name(md.split_score_func) << "\n";
    o << "grow policy: " << policy_name(md.grow_policy) << " verbose: " << md.verbose << " device: " << device_name << "use cv: "
      << static_cast<int>(md.use_cv) << " batch size: " << md.batch_size << "\n";
    o << "Ensemble with: " << md.n_leaves << " leaves from " << md.n_trees << " trees\n";
    o << "Model has: " << static_cast<int>(m.opts.size()) << " optimizers \n";
    return std::move(o.s);
}

}  // namespace gbrl
