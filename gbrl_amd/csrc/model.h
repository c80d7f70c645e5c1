// model.h -- host-side ensemble model (structure-of-arrays) and the .gbrl_model file format.
//
// Mirrors the DATA MODEL of the reference (ensembleMetaData / ensembleData, gbrl/src/cpp/types.h:218-304) and its
// binary serialisation (GBRL::saveToFile/loadFromFile gbrl.cpp:1130-1250, save/load_ensemble_data types.cpp:681-844,
// optimizer/scheduler records optimizer.cpp:120-131, scheduler.cpp:64-108) so that files written here load in the
// reference and vice versa.  Storage is std::vector (grown per tree) instead of the reference's fixed
// 50 000-tree arena; the capacity fields of the metadata are still maintained the way the reference's CPU
// path would (types.h:49-52, types.cpp:847-855) because they are part of the file bytes.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gbrl_hip.h"

namespace gbrl {

constexpr int kCat = GBRL_HIP_MAX_CHAR_SIZE;
constexpr int kInitialMaxTrees = 50000;  // INITAL_MAX_TREES, types.h:49
constexpr int kTreesBatch = 25000;       // TREES_BATCH, types.h:52

static_assert(sizeof(gbrl_hip_metadata) == 80, "ensembleMetaData must be 80 bytes (SURVEY.md A12)");

struct Model {
    gbrl_hip_metadata meta{};
    std::string learner_name = "GBRL";
    bool parallel_predict = true;   // gbrl.h:511 (the reference clears it for Adam only, which this build refuses); saved and loaded with the file

    // ensembleData (types.h:279-304).  S = trees (oblivious) | leaves (greedy)
    std::vector<float> bias, feature_weights;                 // [D], [in]
    std::vector<int32_t> tree_indices;                        // [T] first leaf of tree
    std::vector<int32_t> depths;                              // [S]
    std::vector<float> values;                                // [L*D]
    std::vector<int32_t> feature_indices;                     // [S*md]
    std::vector<float> feature_values;                        // [S*md]
    std::vector<float> edge_weights;                          // [L*md]
    std::vector<uint8_t> is_numerics;                         // [S*md]
    std::vector<uint8_t> inequality_directions;               // [L*md]
    std::vector<char> categorical_values;                     // [S*md*128]
    std::vector<int32_t> feature_mapping, reverse_num, reverse_cat;  // [in]
    std::vector<uint8_t> mapping_numerics;                    // [in]

    std::vector<gbrl_hip_optimizer> opts;

    uint64_t version = 0;  // bumped on every mutation; device mirrors compare against it

    explicit Model(const gbrl_hip_config &cfg);
    Model() = default;

    bool oblivious() const { return meta.grow_policy == GBRL_HIP_GROW_OBLIVIOUS; }
    size_t split_rows() const { return oblivious() ? meta.n_trees : meta.n_leaves; }

    // reserve room for one more tree with `n_leaves_new` leaves; maintains the reference's capacity book-keeping
    void begin_tree();
    void set_feature_mapping(const int32_t *mapping, const uint8_t *is_numeric);
    void add_optimizer(const gbrl_hip_optimizer &o);  // throws std::runtime_error like GBRL::set_optimizer

    void save(const std::string &filename) const;    // throws on I/O error
    static Model load(const std::string &filename);  // throws on I/O error
};

}  // namespace gbrl
