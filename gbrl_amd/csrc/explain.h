// explain.h -- model inspection served from the host copy of the ensemble: SHAP values,
// the C-header export, and the text dumps.  None of this is on the accelerated step/predict path; it exists so that the
// unchanged gbrl Python package finds every method of the reference's gbrl_cpp.GBRL class.
#pragma once

#include <cmath>
#include <string>
#include <vector>

#include "model.h"

namespace gbrl {

// One ensemble member as an explicit binary tree, nodes numbered depth-first (left before right), with everything Linear TreeSHAP
// needs per node (alloc_shap_data shap.cpp:38-168).  Built on the host; evaluated on the host (explain.cpp) or by k_shap (shap.hip).
struct ShapNode {
    int parent = -1, left = -1, right = -1;
    int feature = -1;           // raw condition feature index; -1 for a leaf
    int tied_to = -1;           // parent index when the parent's feature already occurred above it, else -1
    int n_unique = 0;           // max over the leaves below of the number of distinct feature indices on their path
    bool numeric = true;
    float threshold = INFINITY;
    int cond = -1;              // categorical condition: index of its 128-byte value in Model::categorical_values (row*max_depth+depth)
    float weight = 1.0f;        // edge weight from the parent (cumulated with the parent's when tied)
    int pred = -1;              // leaves: offset of the leaf's [D] cover-weighted value in ShapTree::leaf_value
};
struct ShapTree {
    std::vector<ShapNode> nodes;
    std::vector<float> leaf_value;
};
ShapTree build_shap_tree(const Model &m, int tree_idx);   // throws "Invalid tree index"

// argument checks shared by the host and device evaluations: throws "Invalid tree index" / missing-input errors
void check_shap_arguments(const Model &m, int tree_idx, const float *obs, const char *cat_obs, const float *norm_values,
                          const float *base_poly, const float *offset);

// Linear TreeSHAP of ONE tree, accumulated into `out` [n_samples][n_num + n_cat][D] (caller zero-fills).
// Semantics of GBRL::tree_shap / get_shap_values (gbrl.cpp:1269-1303, shap.cpp:38-364).  Host pointers only, like the
// reference's binding (binding.cpp:985-1052 casts to NumPy arrays).
//   norm_values [(max_depth+1)][max_depth], base_poly [max_depth], offset [max_depth][max_depth]  (gbrl/common/utils.py:317-371)
void tree_shap(const Model &m, int tree_idx, const float *obs, const char *cat_obs, int n_samples, const float *norm_values,
               const float *base_poly, const float *offset, float *out);
// every tree of the ensemble, summed (GBRL::ensemble_shap gbrl.cpp:1305-1342)
void ensemble_shap(const Model &m, const float *obs, const char *cat_obs, int n_samples, const float *norm_values,
                   const float *base_poly, const float *offset, float *out);

// Text of the C header GBRL::exportModel writes (gbrl.cpp:1106-1128, export_ensemble_data types.cpp:409-679).
// export_format: "float" | "fxp8" | "fxp16"; export_type: "full" | "compact".  Throws std::runtime_error with the reference's
// messages; returns false (nothing to write, status 0) in the one case the reference returns silently: a compact export of a
// model deeper than 6.
bool export_header(const Model &m, const std::string &model_name, const std::string &export_format, const std::string &export_type,
                   const std::string &prefix, std::string &text);

// stdout text of GBRL::print_tree (gbrl.cpp:1357-1391 + print_leaf node.cpp:492-553); tree_idx -1 = last tree
std::string tree_text(const Model &m, int tree_idx);
// stdout text of GBRL::print_ensemble_metadata (gbrl.cpp:1254-1267)
std::string metadata_text(const Model &m, const char *device_name);

// ensembleData::alloc_data_size as the reference's CPU path would report it.  compact = false: the arena sized by the capacity
// fields max_trees / max_leaves (ensemble_data_alloc types.cpp:194-256), printed inside the export header's comment block;
// compact = true: the exact-size copy get_ensemble_data() hands out (copy_ensemble_data types.cpp:322-384), whose size is what
// get_ensemble_data()["alloc_data_size"] shows (binding.cpp:382)
size_t reference_alloc_bytes(const Model &m, bool compact);

}  // namespace gbrl
