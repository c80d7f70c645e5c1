// TEST INFRASTRUCTURE ONLY.  Thin C entry points onto the REFERENCE's own per-candidate scoring functions (compiled in oracle/_ref from the
// sources under /root/reference): what the near-tie replay of the product has to reproduce bit for bit.
#include <cmath>
#include <cstring>
#include "node.h"
#include "split_candidate_generator.h"
extern "C" {
// TreeNode::splitScoreCosine / splitScoreL2 (node.cpp:187-251, 321-376) of one numeric candidate on the node `rows`
float ref_split_score(const float *obs, const float *grads, const int *rows, int n, int F, int D, int feature, float value, int min_data, int cosine) {
    int *own = new int[n > 0 ? n : 1];
    std::memcpy(own, rows, sizeof(int) * n);
    TreeNode node(own, n, F, 0, D, 0, 0);
    splitCandidate c{feature, value, nullptr};
    return cosine ? node.splitScoreCosine(obs, grads, c, min_data) : node.splitScoreL2(obs, grads, c, min_data);
}
// scoreCosine / scoreL2 (split_candidate_generator.cpp:262-320): the parent score
float ref_parent_score(const float *grads, const int *rows, int n, int D, int cosine) {
    return cosine ? scoreCosine(rows, n, grads, D) : scoreL2(rows, n, grads, D);
}
}
extern "C" {
// TreeNode::splitScoreCosineCategorical / splitScoreL2Categorical (node.cpp:253-319, 378-434): cat_obs [N][Fc] cells of 128 bytes
float ref_split_score_cat(const char *cat_obs, const float *grads, const int *rows, int n, int Fc, int D, int feature, const char *value, int min_data, int cosine) {
    int *own = new int[n > 0 ? n : 1];
    std::memcpy(own, rows, sizeof(int) * n);
    TreeNode node(own, n, 0, Fc, D, 0, 0);
    char cell[128];
    std::memcpy(cell, value, 128);
    splitCandidate c{feature, INFINITY, cell};
    return cosine ? node.splitScoreCosineCategorical(cat_obs, grads, c, min_data) : node.splitScoreL2Categorical(cat_obs, grads, c, min_data);
}
}
