/* oracle/oracle.h -- C API of the CPU restatement (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker, not the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.  The product (libgbrl_hip.so) never
 * links, loads or calls anything declared here.
 *
 * Every function restates -- in this repo's own words -- the reference's CPU path
 * (gbrl/src/cpp, v1.1.6).  Citations are /root/reference-relative file:line.
 */
#ifndef GBRL_ORACLE_H
#define GBRL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_model oracle_model;

/* enum values follow gbrl/src/cpp/types.h:110-181 */
enum { ORACLE_L2 = 0, ORACLE_COSINE = 1 };        /* scoreFunc      */
enum { ORACLE_UNIFORM = 0, ORACLE_QUANTILE = 1 }; /* generatorType  */
enum { ORACLE_GREEDY = 0, ORACLE_OBLIVIOUS = 1 }; /* growPolicy     */

/* GBRL::GBRL (gbrl.cpp:76-114) minus the parts that are out of scope (control variates) */
oracle_model *oracle_create(int input_dim, int output_dim, int max_depth, int min_data_in_leaf,
                            int n_bins, int par_th, int split_score_func, int generator_type,
                            int grow_policy);
void oracle_destroy(oracle_model *m);

/* GBRL::set_bias / set_feature_weights / set_feature_mapping (gbrl.cpp:213-316) */
void oracle_set_bias(oracle_model *m, const float *bias);
void oracle_set_feature_weights(oracle_model *m, const float *w);
void oracle_set_feature_mapping(oracle_model *m, const int32_t *mapping, const uint8_t *is_numeric);
/* GBRL::set_optimizer (gbrl.cpp:452-525), SGD + Const scheduler only */
int oracle_add_sgd(oracle_model *m, float lr, int start_idx, int stop_idx);

/* GBRL::step -> Fitter::step_cpu (gbrl.cpp:939-981, fitter.cpp:50-115) */
int oracle_step(oracle_model *m, const float *obs, const char *cat_obs, const float *grads,
                int n_samples, int n_num_features, int n_cat_features);

/* GBRL::predict -> Predictor::predict_cpu (gbrl.cpp:369-422, predictor.cpp:122-185) */
int oracle_predict(oracle_model *m, const float *obs, const char *cat_obs, int n_samples,
                   int n_num_features, int n_cat_features, int start_tree, int stop_tree,
                   float *preds_out);

/* sizes: [0]=n_trees [1]=n_leaves [2]=split_rows (trees if oblivious else leaves) [3]=max_depth
 *        [4]=output_dim [5]=iteration [6]=n_num_features [7]=n_cat_features */
void oracle_sizes(const oracle_model *m, int32_t out[8]);
/* copy-out of the ensemble arrays (ensembleData, types.h:279-304); any pointer may be NULL */
void oracle_get_ensemble(const oracle_model *m, int32_t *tree_indices, int32_t *depths, float *values,
                         int32_t *feature_indices, float *feature_values, float *edge_weights,
                         uint8_t *is_numerics, uint8_t *inequality_directions, char *categorical_values);

/* split candidates of the LAST step (split_candidate_generator.cpp:59-163): returns n_candidates;
 * feature_idx/value/is_cat sized n_bins*(input_dim); cat sized that *128 */
int oracle_last_candidates(const oracle_model *m, int32_t *feature_idx, float *value, uint8_t *is_cat,
                           char *cat);

#ifdef __cplusplus
}
#endif
#endif
