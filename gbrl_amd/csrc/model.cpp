// model.cpp -- see model.h.  Host only; no HIP.
#include "model.h"

#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace gbrl {

namespace {

// serializationHeader, gbrl/src/cpp/types.h:312-318: u16 major, minor, patch @0,2,4; u64 @8; u32 @16; 24 bytes.
struct FileHeader {
    uint16_t major_version, minor_version, patch_version;
    uint16_t pad0;
    uint64_t reserved1;
    uint32_t reserved2;
    uint32_t pad1;
};
static_assert(sizeof(FileHeader) == 24, "serializationHeader is 24 bytes");
constexpr uint16_t kMajor = 1, kMinor = 1, kPatch = 6;  // gbrl/src/cpp/config.h (v1.1.6)

template <typename T>
void put(std::ofstream &f, const T &v) { f.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
template <typename T>
void get(std::ifstream &f, T &v) { f.read(reinterpret_cast<char *>(&v), sizeof(T)); }

// one "NULL_CHECK + payload" record (types.cpp:696-767); arrays are never null here => always VALID (1)
template <typename T>
void put_array(std::ofstream &f, const std::vector<T> &v, size_t count) {
    const uint8_t valid = 1;
    put(f, valid);
    if (count > v.size()) throw std::runtime_error("internal: short ensemble array");
    f.write(reinterpret_cast<const char *>(v.data()), count * sizeof(T));
}
template <typename T>
void get_array(std::ifstream &f, std::vector<T> &v, size_t count) {
    uint8_t check = 0;
    get(f, check);
    if (check == 1) {
        // the payload must fit into what is left of the file: a corrupt header must not drive a multi-GB allocation
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streampos end = f.tellg();
        f.seekg(here);
        if (!f.good() || static_cast<double>(count) * sizeof(T) > static_cast<double>(end - here)) throw std::runtime_error("Reading file error");
    } else if (static_cast<double>(count) * sizeof(T) > 4e9) {
        throw std::runtime_error("Reading file error");
    }
    v.assign(count, T());
    if (check == 1) f.read(reinterpret_cast<char *>(v.data()), count * sizeof(T));
}

}  // namespace

Model::Model(const gbrl_hip_config &c) {
    // ensemble_metadata_alloc, types.cpp:166-192, called as in GBRL::GBRL gbrl.cpp:82
    meta.n_leaves = 0; meta.n_trees = 0;
    meta.max_trees = kInitialMaxTrees;
    meta.max_leaves = kInitialMaxTrees * (1 << c.max_depth);
    meta.max_trees_batch = kTreesBatch;
    meta.max_leaves_batch = kTreesBatch * (1 << c.max_depth);
    meta.input_dim = c.input_dim; meta.output_dim = c.output_dim; meta.policy_dim = c.policy_dim;
    meta.max_depth = c.max_depth; meta.min_data_in_leaf = c.min_data_in_leaf; meta.n_bins = c.n_bins;
    meta.par_th = c.par_th; meta.cv_beta = c.cv_beta; meta.verbose = c.verbose; meta.batch_size = c.batch_size;
    meta.use_cv = 0;
    meta.split_score_func = static_cast<uint8_t>(c.split_score_func);
    meta.generator_type = static_cast<uint8_t>(c.generator_type);
    meta.grow_policy = static_cast<uint8_t>(c.grow_policy);
    meta.n_num_features = 0; meta.n_cat_features = 0; meta.iteration = 0;
    learner_name = c.learner_name ? c.learner_name : "GBRL";
    // ensemble_data_alloc zero-fills everything (types.cpp:194-262)
    bias.assign(c.output_dim, 0.0f);
    feature_weights.assign(c.input_dim, 0.0f);
    feature_mapping.assign(c.input_dim, 0);
    reverse_num.assign(c.input_dim, 0);
    reverse_cat.assign(c.input_dim, 0);
    mapping_numerics.assign(c.input_dim, 0);
}

void Model::begin_tree() {
    // capacity book-keeping of allocate_ensemble_memory (types.cpp:847-855)
    if (meta.n_leaves >= meta.max_leaves || meta.n_trees >= meta.max_trees) {
        meta.max_leaves = meta.n_leaves + meta.max_leaves_batch;
        meta.max_trees = meta.n_trees + meta.max_trees_batch;
    }
}

void Model::set_feature_mapping(const int32_t *mapping, const uint8_t *is_numeric) {
    // GBRL::set_feature_mapping, gbrl.cpp:271-316: reverse maps are -1 padded
    const int in = meta.input_dim;
    int j = 0, k = 0;
    for (int i = 0; i < in; ++i) { reverse_num[i] = -1; reverse_cat[i] = -1; }
    for (int i = 0; i < in; ++i) {
        feature_mapping[i] = mapping[i];
        mapping_numerics[i] = is_numeric[i] ? 1 : 0;
        if (is_numeric[i]) reverse_num[j++] = i; else reverse_cat[k++] = i;
    }
    ++version;
}

void Model::add_optimizer(const gbrl_hip_optimizer &o) {
    // GBRL::set_optimizer, gbrl.cpp:452-525 (same checks, same order)
    if (opts.size() >= static_cast<size_t>(meta.output_dim)) throw std::runtime_error("Optimizer Limit Reached");
    if (o.start_idx >= o.stop_idx) throw std::runtime_error("invalid index ranges");
    if (o.start_idx < 0 || o.stop_idx <= 0 || o.start_idx >= meta.output_dim || o.stop_idx > meta.output_dim)
        throw std::runtime_error("invalid index ranges");
    if (o.algo != GBRL_HIP_ALGO_SGD)  // the reference's GPU path rejects Adam as well (gbrl.cpp:477-482)
        throw std::runtime_error("Incompatible GPU optimizer");
    if (o.scheduler != GBRL_HIP_SCHED_CONST)  // ... and the Linear scheduler (gbrl.cpp:499-505)
        throw std::runtime_error("Incompatible GPU scheduler");
    opts.push_back(o);
    ++version;
}

void Model::save(const std::string &filename) const {
    std::ofstream f(filename, std::ios::binary);
    if (!f.is_open() || f.fail()) throw std::runtime_error("File opening error");
    FileHeader h{};  // zero padding (the reference leaves its padding bytes uninitialised)
    h.major_version = kMajor; h.minor_version = kMinor; h.patch_version = kPatch;
    put(f, h);
    put(f, meta);
    put(f, static_cast<char>(parallel_predict));
    put(f, static_cast<char>(meta.use_cv));
    const uint64_t name_len = learner_name.size();
    put(f, name_len);
    f.write(learner_name.data(), name_len);
    // save_ensemble_data, types.cpp:681-767 -- 15 records in this order
    const size_t T = meta.n_trees, L = meta.n_leaves, S = split_rows(), md = meta.max_depth;
    const size_t D = meta.output_dim, in = meta.input_dim;
    put_array(f, bias, D);
    put_array(f, feature_weights, in);
    put_array(f, tree_indices, T);
    put_array(f, depths, S);
    put_array(f, values, L * D);
    put_array(f, feature_indices, S * md);
    put_array(f, feature_values, S * md);
    put_array(f, edge_weights, L * md);
    put_array(f, reverse_num, in);
    put_array(f, reverse_cat, in);
    put_array(f, feature_mapping, in);
    put_array(f, mapping_numerics, in);
    put_array(f, is_numerics, S * md);
    put_array(f, inequality_directions, L * md);
    put_array(f, categorical_values, S * md * kCat);
    const int32_t n_opts = static_cast<int32_t>(opts.size());
    put(f, n_opts);
    for (const auto &o : opts) {
        // SGDOptimizer::saveToFile optimizer.cpp:120-131 + ConstScheduler::saveToFile scheduler.cpp:99-108
        put(f, static_cast<uint8_t>(o.algo));
        put(f, static_cast<int32_t>(o.start_idx));
        put(f, static_cast<int32_t>(o.stop_idx));
        put(f, static_cast<uint8_t>(o.scheduler));
        put(f, o.init_lr);
    }
    if (!f.good()) throw std::runtime_error("Writing to file error");
}

Model Model::load(const std::string &filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f.is_open() || f.fail()) throw std::runtime_error("Error opening file");
    Model m;
    FileHeader h{};
    get(f, h);
    if (!f.good()) throw std::runtime_error("Failed to read header from file");
    get(f, m.meta);
    char byte = 0;
    get(f, byte); m.parallel_predict = byte != 0;
    get(f, byte); m.meta.use_cv = byte != 0;
    uint64_t name_len = 0;
    get(f, name_len);
    if (!f.good() || name_len > (1u << 20)) throw std::runtime_error("Reading file error");
    m.learner_name.resize(name_len);
    f.read(&m.learner_name[0], name_len);
    if (!f.good()) throw std::runtime_error("Reading file error");
    const gbrl_hip_metadata &md_ = m.meta;
    if (md_.n_trees < 0 || md_.n_leaves < 0 || md_.max_depth < 0 || md_.max_depth > 30 || md_.output_dim <= 0 ||
        md_.input_dim <= 0)
        throw std::runtime_error("Reading file error");
    const size_t T = md_.n_trees, L = md_.n_leaves, S = m.split_rows(), md = md_.max_depth;
    const size_t D = md_.output_dim, in = md_.input_dim;
    get_array(f, m.bias, D);
    get_array(f, m.feature_weights, in);
    get_array(f, m.tree_indices, T);
    get_array(f, m.depths, S);
    get_array(f, m.values, L * D);
    get_array(f, m.feature_indices, S * md);
    get_array(f, m.feature_values, S * md);
    get_array(f, m.edge_weights, L * md);
    get_array(f, m.reverse_num, in);
    get_array(f, m.reverse_cat, in);
    get_array(f, m.feature_mapping, in);
    get_array(f, m.mapping_numerics, in);
    get_array(f, m.is_numerics, S * md);
    get_array(f, m.inequality_directions, L * md);
    get_array(f, m.categorical_values, S * md * kCat);
    // The predict kernels index with these arrays: a corrupt or hand-edited file must fail here, not fault on the device.
    if (md_.n_num_features < 0 || md_.n_cat_features < 0 || md_.n_num_features + md_.n_cat_features > md_.input_dim) throw std::runtime_error("Reading file error");
    for (size_t t = 0; t < T; ++t) {
        const int32_t a = m.tree_indices[t], b = t + 1 < T ? m.tree_indices[t + 1] : static_cast<int32_t>(L);
        if (a < 0 || a > b || b > static_cast<int32_t>(L) || (t == 0 && a != 0)) throw std::runtime_error("Reading file error");
    }
    for (size_t r = 0; r < S; ++r) {
        if (m.depths[r] < 0 || m.depths[r] > static_cast<int32_t>(md)) throw std::runtime_error("Reading file error");
        for (int32_t d = 0; d < m.depths[r]; ++d) {
            const int32_t fi = m.feature_indices[r * md + d];
            const int32_t lim = m.is_numerics[r * md + d] ? md_.n_num_features : md_.n_cat_features;
            if (fi < 0 || fi >= lim) throw std::runtime_error("Reading file error");
        }
    }
    if (md_.grow_policy == GBRL_HIP_GROW_OBLIVIOUS)
        for (size_t t = 0; t < T; ++t) {   // an oblivious tree owns 2^depth leaves
            const int64_t leaves = (t + 1 < T ? m.tree_indices[t + 1] : static_cast<int32_t>(L)) - m.tree_indices[t];
            if (leaves != (int64_t(1) << m.depths[t])) throw std::runtime_error("Reading file error");
        }
    int32_t n_opts = 0;
    get(f, n_opts);
    if (!f.good() || n_opts < 0 || n_opts > md_.output_dim) throw std::runtime_error("Optimizer load error");
    for (int i = 0; i < n_opts; ++i) {
        gbrl_hip_optimizer o{};
        uint8_t algo = 0, sched = 0;
        int32_t start = 0, stop = 0;
        get(f, algo); get(f, start); get(f, stop);
        o.beta_1 = 0.9f; o.beta_2 = 0.999f; o.eps = 1e-8f; o.stop_lr = 1e-8f; o.T = 10000;
        if (algo == GBRL_HIP_ALGO_ADAM) {  // AdamOptimizer::saveToFile optimizer.cpp:223-237: 3 floats after stop
            get(f, o.beta_1); get(f, o.beta_2); get(f, o.eps);
        }
        get(f, sched);
        get(f, o.init_lr);
        if (sched == GBRL_HIP_SCHED_LINEAR) { get(f, o.stop_lr); get(f, o.T); }  // scheduler.cpp:64-75
        o.algo = algo; o.scheduler = sched; o.start_idx = start; o.stop_idx = stop;
        if (!f.good()) throw std::runtime_error("Optimizer load error");
        // Files with Adam / Linear records parse, but this build cannot run them (no CPU path): refuse loudly.
        if (algo != GBRL_HIP_ALGO_SGD || sched != GBRL_HIP_SCHED_CONST)
            throw std::runtime_error("model file uses Adam or a Linear scheduler: CPU-only in the reference, unsupported here");
        m.opts.push_back(o);
    }
    return m;
}

}  // namespace gbrl
