// oracle/oracle.cpp -- CPU restatement of GBRL's per-step tree fit and ensemble predict.
//
// TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP product in gbrl_amd/csrc:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
// The product never links or calls it.
//
// Parity status: PINNED.  tests/test_oracle.py checks this restatement (a) against the committed
// golden vectors in tests/golden/*.npz, which were produced by the reference's own CPU path
// compiled from /root/reference (oracle/Makefile target `ref`, script tests/golden/make_golden.py),
// and (b) live against oracle/_ref whenever that build is present.
//
// What is restated (all citations are /root/reference-relative):
//   step      Fitter::step_cpu                         gbrl/src/cpp/fitter.cpp:50-115
//   grads     calculate_mean / calculate_std_and_center / divide_mat_by_vec_inplace
//                                                      gbrl/src/cpp/math_ops.cpp:255-300, 461-513, 79-105
//   cands     uniform / quantile / categorical         gbrl/src/cpp/split_candidate_generator.cpp:59-163, 216-249
//   scoring   TreeNode::getSplitScore and friends      gbrl/src/cpp/node.cpp:151-434, math_ops.h:432-575
//   growth    fit_greedy_tree / fit_oblivious_tree     gbrl/src/cpp/fitter.cpp:263-484
//   leaves    calc_leaf_value                          gbrl/src/cpp/fitter.cpp:545-582
//   predict   predict_cpu / predict_over_trees / predict_over_leaves / SGDOptimizer::step
//                                                      gbrl/src/cpp/predictor.cpp:122-265, optimizer.cpp:110-118
// The algorithm is the reference's brute-force one on purpose (per-candidate scan of the node's rows,
// float32 sums in row order); the split and parent scores are evaluated by replay_sequence.cpp, the reference's operation sequence
// spelled out (explicit fmaf, no contraction), so that their bits do not depend on this file's compiler; it is also what bench.py times as the "port" CPU baseline.
//
// Deliberately NOT restated: control variates, Adam / Linear scheduler, fit(), SHAP, export.

#include "oracle.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

// the reference's float32 operation sequence, spelled out (replay_sequence.cpp, built without contraction; pinned bit for bit to the
// reference's own functions by tests/test_oracle.py)
extern "C" float oracle_seq_split(const float *g, const int *ridx, int nr, const int *lidx, int nl, int D, int cosine);
extern "C" float oracle_replay_parent_cosine(const float *g, const int *rows, int n, int D);
extern "C" float oracle_replay_parent_l2(const float *g, const int *rows, int n, int D);

namespace {

constexpr int kCat = 128;  // MAX_CHAR_SIZE, types.h:56

struct Cond {           // splitCondition, types.h:64-70
    int feat = 0;
    float value = 0.f;
    bool dir = false;
    float edge_w = 0.f;
    bool is_cat = false;
    char cat[kCat] = {0};
};

struct Candidate {      // splitCandidate, types.h:76-80
    int feat = 0;
    float value = 0.f;
    bool is_cat = false;
    char cat[kCat] = {0};
};

struct Sgd { float lr; int start, stop; };

struct Node {           // TreeNode, node.h
    std::vector<int> rows;
    int depth = 0;
    std::vector<Cond> path;
};

}  // namespace

struct oracle_model {
    int input_dim, output_dim, max_depth, min_data_in_leaf, n_bins, par_th;
    int score, gen, policy;
    int n_num = 0, n_cat = 0, iteration = 0, n_trees = 0, n_leaves = 0;
    std::vector<float> bias, feature_weights;
    std::vector<int> feature_mapping, rev_num, rev_cat;
    std::vector<uint8_t> mapping_numerics;
    // ensembleData, types.h:279-304 (grown on demand instead of the 50 000-tree arena)
    std::vector<int> tree_indices, depths, feature_indices;
    std::vector<float> values, feature_values, edge_weights;
    std::vector<uint8_t> is_numerics, inequality_directions;
    std::vector<char> categorical_values;
    std::vector<Sgd> opts;
    std::vector<Candidate> last_candidates;
    int max_threads = 1;  // omp_get_max_threads() at creation; restored before every call
};

namespace {

// utils.h:64-81
int calc_threads(int total, int min_per_thread) {
    int max_threads = omp_get_max_threads();
    int n = total / min_per_thread;
    if (n > total) n = total;
    if (n <= 1) return 1;
    return n > max_threads ? max_threads : n;
}

// math_ops.cpp:255-300 -- per-thread partial sums, combined in thread order
std::vector<float> column_mean(const float *mat, int n_samples, int n_cols, int par_th) {
    const int n_el = n_samples * n_cols;
    std::vector<float> mean(n_cols, 0.0f);
    const float recip = 1.0f / static_cast<float>(n_samples);
    const int nt = calc_threads(n_el, par_th);
    if (nt > 1) {
        omp_set_num_threads(nt);
        const int per = n_el / nt;
        std::vector<float> part(static_cast<size_t>(nt) * n_cols, 0.0f);
#pragma omp parallel
        {
            const int t = omp_get_thread_num();
            const int lo = t * per, hi = (t == nt - 1) ? n_el : lo + per;
            float *p = part.data() + static_cast<size_t>(t) * n_cols;
            for (int i = lo; i < hi; ++i) p[i % n_cols] += mat[i];
        }
        for (int d = 0; d < nt * n_cols; ++d) mean[d % n_cols] += part[d];
    } else {
        for (int i = 0; i < n_el; ++i) mean[i % n_cols] += mat[i];
    }
    for (int d = 0; d < n_cols; ++d) mean[d] *= recip;
    return mean;
}

// math_ops.cpp:461-513 -- centre in place, return unbiased std
std::vector<float> column_std_and_center(float *mat, const float *mean, int n_samples, int n_cols,
                                         int par_th) {
    const int n_el = n_samples * n_cols;
    std::vector<float> var(n_cols, 0.0f);
    const float recip = 1.0f / (static_cast<float>(n_samples) - 1.0f);
    const int nt = calc_threads(n_el, par_th);
    if (nt > 1) {
        omp_set_num_threads(nt);
        const int per = n_el / nt;
        std::vector<float> part(static_cast<size_t>(nt) * n_cols, 0.0f);
#pragma omp parallel
        {
            const int t = omp_get_thread_num();
            const int lo = t * per, hi = (t == nt - 1) ? n_el : lo + per;
            float *p = part.data() + static_cast<size_t>(t) * n_cols;
            for (int i = lo; i < hi; ++i) {
                const int c = i % n_cols;
                const float v = mat[i] - mean[c];
                p[c] += (v * v);
                mat[i] -= mean[c];
            }
        }
        for (int d = 0; d < nt * n_cols; ++d) var[d % n_cols] += part[d];
    } else {
        for (int i = 0; i < n_el; ++i) {
            const int c = i % n_cols;
            const float v = mat[i] - mean[c];
            var[c] += (v * v);
            mat[i] -= mean[c];
        }
    }
    for (int d = 0; d < n_cols; ++d) var[d] = sqrtf(var[d] * recip);
    return var;
}

// math_ops.cpp:79-105
void divide_by_vec(float *mat, const float *vec, int n_samples, int n_cols) {
    const long n_el = static_cast<long>(n_samples) * n_cols;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n_el; ++i) mat[i] /= (vec[i % n_cols] + 1e-8f);
}

// split_candidate_generator.cpp:293-320 (parent score, greedy/L2) and :262-290 + math_ops.h:504-524 (greedy/Cosine): replay_sequence.cpp
float parent_l2(const std::vector<int> &rows, const float *g, int n_cols) {
    return oracle_replay_parent_l2(g, rows.data(), static_cast<int>(rows.size()), n_cols);
}
float parent_cosine(const std::vector<int> &rows, const float *g, int n_cols) {
    if (rows.empty()) return 0.0f;
    return oracle_replay_parent_cosine(g, rows.data(), static_cast<int>(rows.size()), n_cols);
}

inline bool goes_right(const oracle_model *m, const float *obs, const char *cat, int row,
                       const Candidate &c) {
    if (c.is_cat)
        return strcmp(&cat[(static_cast<size_t>(row) * m->n_cat + c.feat) * kCat], c.cat) == 0;
    return obs[static_cast<size_t>(row) * m->n_num + c.feat] > c.value;
}

// node.cpp:151-434 -- one candidate, one node, brute force over the node's rows
float split_score(const oracle_model *m, const Node &node, const float *obs, const char *cat,
                  const float *bg, const Candidate &cand) {
    const int D = m->output_dim;
    for (int i = 0; i < node.depth; ++i) {  // node.cpp:154-166: no re-use along a path
        const Cond &p = node.path[i];
        if (!cand.is_cat) {
            if (!p.is_cat && p.value == cand.value && p.feat == cand.feat) return -INFINITY;
        } else {
            if (p.is_cat && strcmp(p.cat, cand.cat) == 0 && p.feat == cand.feat) return -INFINITY;
        }
    }
    const int n = static_cast<int>(node.rows.size());
    std::vector<int> lidx(n), ridx(n);
    int nl = 0, nr = 0;
    for (int k = 0; k < n; ++k) {
        const int row = node.rows[k];
        if (goes_right(m, obs, cat, row, cand)) ridx[nr++] = row; else lidx[nl++] = row;
    }
    if (nl < m->min_data_in_leaf || nr < m->min_data_in_leaf) return -INFINITY;
    // node.cpp:187-251 / 321-376 + math_ops.h:432-575 as the reference's release build evaluates them (replay_sequence.cpp)
    return oracle_seq_split(bg, ridx.data(), nr, lidx.data(), nl, D, m->score == ORACLE_COSINE ? 1 : 0);
}

// node.cpp:64-149
void split_node(const oracle_model *m, const Node &node, const float *obs, const char *cat,
                const Candidate &c, Node &left, Node &right) {
    left = Node(); right = Node();
    for (int row : node.rows) (goes_right(m, obs, cat, row, c) ? right.rows : left.rows).push_back(row);
    left.depth = right.depth = node.depth + 1;
    left.path = node.path; right.path = node.path;
    const float np = static_cast<float>(node.rows.size());
    Cond cl;
    cl.feat = c.feat; cl.value = c.value; cl.is_cat = c.is_cat;
    memcpy(cl.cat, c.cat, kCat);
    Cond cr = cl;
    cl.dir = false;
    cl.edge_w = node.rows.empty() ? 0.0f : static_cast<float>(left.rows.size()) / np;
    cr.dir = true;
    cr.edge_w = node.rows.empty() ? 0.0f : static_cast<float>(right.rows.size()) / np;
    left.path.push_back(cl);
    right.path.push_back(cr);
}

// split_candidate_generator.cpp:59-76.  NOTE min + b*step is contracted to ONE fused multiply-add
// by the reference's release build (-O3 on an FMA target); it is written fmaf() here so the bits do
// not depend on this file's compiler flags.
void uniform_candidates(oracle_model *m, const float *obs, int n) {
    const int F = m->n_num, B = m->n_bins;
    std::vector<float> mx(F, -INFINITY), mn(F, INFINITY);
    for (int i = 0; i < n; ++i)
        for (int f = 0; f < F; ++f) {
            const float v = obs[static_cast<size_t>(i) * F + f];
            mx[f] = mx[f] > v ? mx[f] : v;
            mn[f] = mn[f] < v ? mn[f] : v;
        }
    for (int f = 0; f < F; ++f) {
        const float step = (mx[f] - mn[f]) / static_cast<float>(B);
        for (int b = 0; b < B; ++b) {
            Candidate c;
            c.feat = f;
            c.value = fmaf(static_cast<float>(b), step, mn[f]);
            m->last_candidates.push_back(c);
        }
    }
}

// fitter.cpp:77-90 + split_candidate_generator.cpp:79-115, 216-249.  The value at a sorted rank does
// not depend on how ties are ordered, so sorting the column's VALUES is equivalent to the reference's
// argsort.  The de-duplication test in computeQuantiles is dead code on the CPU path (Q1): exactly
// n_bins candidates per feature, duplicates kept.
void quantile_candidates(oracle_model *m, const float *obs, int n) {
    const int F = m->n_num, B = m->n_bins;
    const int actual = B + 1, per = n / actual;
    int rem = n % actual;
    std::vector<int> counts(actual, per);
    while (rem > 0)
        for (int i = 0; i < actual; ++i) {
            counts[i] += 1;
            if (--rem == 0) break;
        }
    std::vector<Candidate> out(static_cast<size_t>(F) * B);
#pragma omp parallel
    {
        std::vector<float> col(n);
#pragma omp for schedule(dynamic, 1)
        for (int f = 0; f < F; ++f) {
            for (int i = 0; i < n; ++i) col[i] = obs[static_cast<size_t>(i) * F + f];
            std::sort(col.begin(), col.end());
            int cum = 0;
            for (int b = 0; b < B; ++b) {
                cum += counts[b];
                Candidate c;
                c.feat = f;
                c.value = col[cum - 1];   // cum >= 1 for every n >= 1: the remainder loop hands the first n % (B+1) buckets one row each
                out[static_cast<size_t>(f) * B + b] = c;
            }
        }
    }
    m->last_candidates.insert(m->last_candidates.end(), out.begin(), out.end());
}

// split_candidate_generator.cpp:117-163 (same container, same insertion order => same iteration order, Q8)
struct CatInfo { float total = 0.0f; int count = 0; int feat = 0; std::string name; };

void categorical_candidates(oracle_model *m, const char *cat, const float *norms, int n) {
    const int Fc = m->n_cat;
    std::unordered_map<std::string, CatInfo> uniq;
    for (int f = 0; f < Fc; ++f)
        for (int i = 0; i < n; ++i) {
            std::string name(cat + (static_cast<size_t>(i) * Fc + f) * kCat, kCat);
            std::string key = name + "_" + std::to_string(f);
            CatInfo &ci = uniq[key];
            ci.total += norms[i];
            ci.count += 1;
            ci.feat = f;
            ci.name = name;
        }
    std::vector<std::pair<std::string, float>> vec;
    for (const auto &kv : uniq) vec.emplace_back(kv.first, kv.second.total / kv.second.count);
    int n_unique = static_cast<int>(vec.size());
    if (n_unique > Fc * m->n_bins) {
        std::sort(vec.begin(), vec.end(),
                  [](const std::pair<std::string, float> &a, const std::pair<std::string, float> &b) {
                      return a.second > b.second;
                  });
        n_unique = Fc * m->n_bins;
    }
    for (int i = 0; i < n_unique; ++i) {
        const CatInfo &ci = uniq[vec[i].first];
        Candidate c;
        c.feat = ci.feat;
        c.value = INFINITY;
        c.is_cat = true;
        memcpy(c.cat, ci.name.c_str(), kCat);  // c_str() of a 128-byte std::string: 128 bytes + NUL
        m->last_candidates.push_back(c);
    }
}

void ensure_capacity(oracle_model *m, int extra_leaves) {
    const int md = m->max_depth, D = m->output_dim;
    const size_t leaves = static_cast<size_t>(m->n_leaves) + extra_leaves;
    const size_t trees = static_cast<size_t>(m->n_trees) + 1;
    const size_t splits = (m->policy == ORACLE_OBLIVIOUS) ? trees : leaves;
    m->tree_indices.resize(trees, 0);
    m->depths.resize(splits, 0);
    m->values.resize(leaves * D, 0.0f);
    m->feature_indices.resize(splits * md, 0);
    m->feature_values.resize(splits * md, 0.0f);
    m->edge_weights.resize(leaves * md, 0.0f);
    m->is_numerics.resize(splits * md, 0);
    m->inequality_directions.resize(leaves * md, 0);
    m->categorical_values.resize(splits * md * kCat, 0);
}

void write_conditions(oracle_model *m, const Node &node, size_t split_row, size_t leaf_row) {
    const int md = m->max_depth;
    for (int i = 0; i < node.depth; ++i) {
        const Cond &c = node.path[i];
        if (c.is_cat) {
            memcpy(&m->categorical_values[(split_row * md + i) * kCat], c.cat, kCat);
            m->is_numerics[split_row * md + i] = 0;
        } else {
            m->is_numerics[split_row * md + i] = 1;
        }
        m->feature_indices[split_row * md + i] = c.feat;
        m->feature_values[split_row * md + i] = c.value;
        m->inequality_directions[leaf_row * md + i] = c.dir;
        m->edge_weights[leaf_row * md + i] = c.edge_w;
    }
}

// Lowest index among maxima wins: the reference's per-thread ascending stripes with strict '>' and an
// in-order merge of the per-thread bests are equivalent to one ascending scan (fitter.cpp:318-354, 411-457).
void argmax_lowest(const std::vector<float> &scores, float &best, int &idx) {
    best = -INFINITY;
    idx = 0;
    for (size_t j = 0; j < scores.size(); ++j)
        if (scores[j] > best) { best = scores[j]; idx = static_cast<int>(j); }
}

// fitter.cpp:263-375
int fit_greedy(oracle_model *m, const float *obs, const char *cat, const float *bg, int n) {
    ensure_capacity(m, 0);
    m->tree_indices[m->n_trees] = m->n_leaves;
    const auto &cands = m->last_candidates;
    const int C = static_cast<int>(cands.size());
    std::vector<Node> stack(1);
    stack[0].rows.resize(n);
    std::iota(stack[0].rows.begin(), stack[0].rows.end(), 0);
    int added = 0, chosen = 0;
    std::vector<float> scores(C);
    while (!stack.empty()) {
        Node node = std::move(stack.back());
        stack.pop_back();
        const bool to_split = !(node.depth == m->max_depth || node.rows.empty() || C == 0);
        float best = -INFINITY;
        if (to_split) {
            float parent = (m->score == ORACLE_COSINE) ? parent_cosine(node.rows, bg, m->output_dim)
                                                        : parent_l2(node.rows, bg, m->output_dim);
            if (node.depth == 0) parent = 0.0f;
#pragma omp parallel for schedule(static)
            for (int j = 0; j < C; ++j) {
                float s = split_score(m, node, obs, cat, bg, cands[j]);
                const int fi = cands[j].is_cat ? cands[j].feat + m->n_num : cands[j].feat;  // Q6
                scores[j] = s * m->feature_weights[fi] - parent;
            }
            argmax_lowest(scores, best, chosen);
        }
        if (best >= 0 && to_split) {
            Node l, r;
            split_node(m, node, obs, cat, cands[chosen], l, r);
            stack.push_back(std::move(r));  // right pushed first => left popped first (DFS, left first)
            stack.push_back(std::move(l));
        } else {
            ensure_capacity(m, 1);
            const size_t idx = m->n_leaves;
            m->depths[idx] = node.depth;
            write_conditions(m, node, idx, idx);
            m->n_leaves += 1;
            added += 1;
        }
    }
    m->n_trees += 1;
    return added;
}

// fitter.cpp:377-484
int fit_oblivious(oracle_model *m, const float *obs, const char *cat, const float *bg, int n) {
    ensure_capacity(m, 0);
    m->tree_indices[m->n_trees] = m->n_leaves;
    const auto &cands = m->last_candidates;
    const int C = static_cast<int>(cands.size());
    std::vector<Node> level(1);
    level[0].rows.resize(n);
    std::iota(level[0].rows.begin(), level[0].rows.end(), 0);
    int depth = 0, chosen = 0;
    std::vector<float> scores(C);
    while (depth < m->max_depth) {
        float best;
#pragma omp parallel for schedule(static)
        for (int j = 0; j < C; ++j) {
            float s = 0.0f;
            for (const Node &node : level) s += split_score(m, node, obs, cat, bg, cands[j]);
            const int fi = cands[j].is_cat ? m->rev_cat[cands[j].feat] : m->rev_num[cands[j].feat];
            scores[j] = s * m->feature_weights[fi];
        }
        argmax_lowest(scores, best, chosen);
        if (best == -INFINITY) break;
        std::vector<Node> next(level.size() * 2);
        for (size_t k = 0; k < level.size(); ++k)
            split_node(m, level[k], obs, cat, cands[chosen], next[2 * k], next[2 * k + 1]);
        level.swap(next);
        depth += 1;
    }
    const int n_nodes = 1 << depth;
    ensure_capacity(m, n_nodes);
    const size_t tree = m->n_trees;
    for (int k = 0; k < n_nodes; ++k) {
        m->depths[tree] = level[k].depth;
        write_conditions(m, level[k], tree, m->n_leaves);
        m->n_leaves += 1;
    }
    m->n_trees += 1;
    return n_nodes;
}

// fitter.cpp:545-582 -- mean of RAW grads over the rows that satisfy the leaf's path; a depth-0
// leaf never "passes" and keeps value 0 (Q7)
void leaf_value(oracle_model *m, const float *obs, const char *cat, const float *grads, int n,
                int leaf_idx, int tree_idx) {
    const int D = m->output_dim, md = m->max_depth;
    const bool obl = (m->policy == ORACLE_OBLIVIOUS);
    const int depth = obl ? m->depths[tree_idx] : m->depths[leaf_idx];
    const size_t cond = static_cast<size_t>(obl ? tree_idx : leaf_idx) * md;
    const size_t ineq = static_cast<size_t>(leaf_idx) * md;
    float count = 0;
    float *val = &m->values[static_cast<size_t>(leaf_idx) * D];
    for (int i = 0; i < n; ++i) {
        bool passed = false;
        for (int d = depth - 1; d >= 0; --d) {
            const int f = m->feature_indices[cond + d];
            const bool test = m->is_numerics[cond + d]
                ? (obs[static_cast<size_t>(i) * m->n_num + f] > m->feature_values[cond + d])
                : (strcmp(&cat[(static_cast<size_t>(i) * m->n_cat + f) * kCat],
                          &m->categorical_values[(cond + d) * kCat]) == 0);
            passed = (test == static_cast<bool>(m->inequality_directions[ineq + d]));
            if (!passed) break;
        }
        if (passed) {
            for (int d = 0; d < D; ++d) val[d] += grads[static_cast<size_t>(i) * D + d];
            count += 1;
        }
    }
    if (count > 0)
        for (int d = 0; d < D; ++d) val[d] /= count;
}

inline void sgd_step(const oracle_model *m, float *theta, const float *leaf) {  // optimizer.cpp:110-118
    for (const Sgd &o : m->opts)
        for (int i = o.start; i < o.stop; ++i) theta[i] -= o.lr * leaf[i];
}

// predictor.cpp:231-265
void predict_row_oblivious(const oracle_model *m, const float *obs, const char *cat, float *theta,
                           int row, int start, int stop) {
    const int md = m->max_depth, D = m->output_dim;
    for (int t = start; t < stop; ++t) {
        const size_t cond = static_cast<size_t>(t) * md;
        int leaf = 0;
        const int depth = m->depths[t];
        for (int d = 0; d < depth; ++d) {
            const int f = m->feature_indices[cond + d];
            const bool passed = m->is_numerics[cond + d]
                ? (obs[static_cast<size_t>(row) * m->n_num + f] > m->feature_values[cond + d])
                : (strcmp(&cat[(static_cast<size_t>(row) * m->n_cat + f) * kCat],
                          &m->categorical_values[(cond + d) * kCat]) == 0);
            leaf |= (passed << (depth - 1 - d));
        }
        sgd_step(m, theta + static_cast<size_t>(row) * D,
                 &m->values[static_cast<size_t>(m->tree_indices[t] + leaf) * D]);
    }
}

// predictor.cpp:188-229 -- walks leaves in order; a leaf that never passes (depth 0) is walked past
// into the NEXT tree's leaves while tree_idx stays put (Q7) -- reproduced as is.
void predict_row_greedy(const oracle_model *m, const float *obs, const char *cat, float *theta,
                        int row, int start, int stop) {
    const int md = m->max_depth, D = m->output_dim;
    int t = start;
    int leaf = m->tree_indices[t];
    while (leaf < m->n_leaves && t < stop) {
        const int depth = m->depths[leaf];
        const size_t cond = static_cast<size_t>(leaf) * md;
        bool passed = false;
        for (int d = depth - 1; d >= 0; --d) {
            const int f = m->feature_indices[cond + d];
            const bool test = m->is_numerics[cond + d]
                ? (obs[static_cast<size_t>(row) * m->n_num + f] > m->feature_values[cond + d])
                : (strcmp(&cat[(static_cast<size_t>(row) * m->n_cat + f) * kCat],
                          &m->categorical_values[(cond + d) * kCat]) == 0);
            passed = (test == static_cast<bool>(m->inequality_directions[cond + d]));
            if (!passed) break;
        }
        if (passed) {
            sgd_step(m, theta + static_cast<size_t>(row) * D, &m->values[static_cast<size_t>(leaf) * D]);
            ++t;
            if (t < stop) leaf = m->tree_indices[t];
        } else {
            ++leaf;
        }
    }
}

}  // namespace

extern "C" {

oracle_model *oracle_create(int input_dim, int output_dim, int max_depth, int min_data_in_leaf,
                            int n_bins, int par_th, int split_score_func, int generator_type,
                            int grow_policy) {
    oracle_model *m = new oracle_model();
    m->max_threads = omp_get_max_threads();
    m->input_dim = input_dim; m->output_dim = output_dim; m->max_depth = max_depth;
    m->min_data_in_leaf = min_data_in_leaf; m->n_bins = n_bins; m->par_th = par_th;
    m->score = split_score_func; m->gen = generator_type; m->policy = grow_policy;
    m->bias.assign(output_dim, 0.0f);
    m->feature_weights.assign(input_dim, 0.0f);  // zero until set, like ensemble_data_alloc (types.cpp:205-207)
    m->feature_mapping.assign(input_dim, 0);
    m->rev_num.assign(input_dim, 0);
    m->rev_cat.assign(input_dim, 0);
    m->mapping_numerics.assign(input_dim, 0);
    return m;
}

void oracle_destroy(oracle_model *m) { delete m; }

void oracle_set_bias(oracle_model *m, const float *b) { m->bias.assign(b, b + m->output_dim); }
void oracle_set_feature_weights(oracle_model *m, const float *w) {
    m->feature_weights.assign(w, w + m->input_dim);
}
void oracle_set_feature_mapping(oracle_model *m, const int32_t *mapping, const uint8_t *is_numeric) {
    int j = 0, k = 0;  // gbrl.cpp:282-296
    for (int i = 0; i < m->input_dim; ++i) {
        m->rev_num[i] = -1;
        m->rev_cat[i] = -1;
        m->feature_mapping[i] = mapping[i];
        m->mapping_numerics[i] = is_numeric[i];
    }
    for (int i = 0; i < m->input_dim; ++i) {
        if (is_numeric[i]) m->rev_num[j++] = i; else m->rev_cat[k++] = i;
    }
}
int oracle_add_sgd(oracle_model *m, float lr, int start_idx, int stop_idx) {
    if (static_cast<int>(m->opts.size()) >= m->output_dim) return -1;
    if (start_idx >= stop_idx || start_idx < 0 || stop_idx > m->output_dim) return -1;
    m->opts.push_back({lr, start_idx, stop_idx});
    return 0;
}

int oracle_step(oracle_model *m, const float *obs, const char *cat_obs, const float *grads,
                int n, int n_num, int n_cat) {
    if (m->iteration == 0) { m->n_num = n_num; m->n_cat = n_cat; }
    if (n_num != m->n_num || n_cat != m->n_cat) return -1;
    const int D = m->output_dim;
    omp_set_num_threads(m->max_threads);
    std::vector<float> bg(grads, grads + static_cast<size_t>(n) * D);  // fitter.cpp:57
    if (m->score == ORACLE_L2) {                                          // fitter.cpp:58-64
        std::vector<float> mean = column_mean(bg.data(), n, D, m->par_th);
        std::vector<float> sd = column_std_and_center(bg.data(), mean.data(), n, D, m->par_th);
        divide_by_vec(bg.data(), sd.data(), n, D);
    }
    std::vector<float> norms;
    if (n_cat > 0) {                                                      // fitter.cpp:66-70
        norms.assign(n, 0.0f);
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < D; ++d)
                norms[i] += grads[static_cast<size_t>(i) * D + d] * grads[static_cast<size_t>(i) * D + d];
    }
    m->last_candidates.clear();
    if (n_num > 0) {
        if (m->gen == ORACLE_UNIFORM) uniform_candidates(m, obs, n);
        else quantile_candidates(m, obs, n);
    }
    if (n_cat > 0) categorical_candidates(m, cat_obs, norms.data(), n);
    omp_set_num_threads(m->max_threads);
    const int added = (m->policy == ORACLE_GREEDY) ? fit_greedy(m, obs, cat_obs, bg.data(), n)
                                                   : fit_oblivious(m, obs, cat_obs, bg.data(), n);
    const int tree = m->n_trees - 1;
    for (int l = 0; l < added; ++l) leaf_value(m, obs, cat_obs, grads, n, m->tree_indices[tree] + l, tree);
    m->iteration += 1;
    return 0;
}

int oracle_predict(oracle_model *m, const float *obs, const char *cat_obs, int n, int n_num,
                   int n_cat, int start_tree, int stop_tree, float *preds) {
    if (m->iteration == 0) { m->n_num = n_num; m->n_cat = n_cat; }
    if (n_num + n_cat != m->input_dim || n_num != m->n_num || n_cat != m->n_cat) return -1;
    const int D = m->output_dim;
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < D; ++d) preds[static_cast<size_t>(i) * D + d] = 0.0f + m->bias[d];
    if (m->n_trees == 0) return 0;
    if (stop_tree > m->n_trees) return -2;
    if (stop_tree == 0) stop_tree = m->n_trees;
    if (m->opts.empty()) return -3;
    omp_set_num_threads(m->max_threads);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        if (m->policy == ORACLE_OBLIVIOUS) predict_row_oblivious(m, obs, cat_obs, preds, i, start_tree, stop_tree);
        else predict_row_greedy(m, obs, cat_obs, preds, i, start_tree, stop_tree);
    }
    return 0;
}

void oracle_sizes(const oracle_model *m, int32_t out[8]) {
    out[0] = m->n_trees; out[1] = m->n_leaves;
    out[2] = (m->policy == ORACLE_OBLIVIOUS) ? m->n_trees : m->n_leaves;
    out[3] = m->max_depth; out[4] = m->output_dim; out[5] = m->iteration;
    out[6] = m->n_num; out[7] = m->n_cat;
}

void oracle_get_ensemble(const oracle_model *m, int32_t *tree_indices, int32_t *depths, float *values,
                         int32_t *feature_indices, float *feature_values, float *edge_weights,
                         uint8_t *is_numerics, uint8_t *inequality_directions, char *categorical_values) {
    const size_t T = m->n_trees, L = m->n_leaves, md = m->max_depth, D = m->output_dim;
    const size_t S = (m->policy == ORACLE_OBLIVIOUS) ? T : L;
    if (tree_indices) memcpy(tree_indices, m->tree_indices.data(), T * 4);
    if (depths) memcpy(depths, m->depths.data(), S * 4);
    if (values) memcpy(values, m->values.data(), L * D * 4);
    if (feature_indices) memcpy(feature_indices, m->feature_indices.data(), S * md * 4);
    if (feature_values) memcpy(feature_values, m->feature_values.data(), S * md * 4);
    if (edge_weights) memcpy(edge_weights, m->edge_weights.data(), L * md * 4);
    if (is_numerics) memcpy(is_numerics, m->is_numerics.data(), S * md);
    if (inequality_directions) memcpy(inequality_directions, m->inequality_directions.data(), L * md);
    if (categorical_values) memcpy(categorical_values, m->categorical_values.data(), S * md * kCat);
}

int oracle_last_candidates(const oracle_model *m, int32_t *feature_idx, float *value, uint8_t *is_cat,
                           char *cat) {
    const int C = static_cast<int>(m->last_candidates.size());
    for (int j = 0; j < C; ++j) {
        const Candidate &c = m->last_candidates[j];
        if (feature_idx) feature_idx[j] = c.feat;
        if (value) value[j] = c.value;
        if (is_cat) is_cat[j] = c.is_cat;
        if (cat) memcpy(cat + static_cast<size_t>(j) * kCat, c.cat, kCat);
    }
    return C;
}

}  // extern "C"
