/* gbrl_hip.h -- C ABI of the MI355X-native GBRL hot path (libgbrl_hip.so).
 *
 * This is the drop-in boundary for ONE path of NVlabs/gbrl: the per-step tree fit (GBRL::step) and the
 * ensemble prediction (GBRL::predict), plus the model state those two calls read and write (constructor,
 * setters/getters, .gbrl_model save/load).  Plain pointers and sizes only; no torch / pybind types.
 * Every entry point names the reference interface it replaces (paths relative to the reference repo).
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md; this repo's own binding
 * (gbrl_amd/csrc/binding.cpp -> Python module `gbrl_cpp`, class `GBRL`) is written against this header only.
 *
 * Conventions
 *   - every function that can fail returns 0 on success and a negative gbrl_hip_status otherwise; the message
 *     is available (thread-local) through gbrl_hip_last_error().  The reference throws std::runtime_error at
 *     the same places; the binding turns a non-zero status back into that exception.
 *   - `*_on_device` flags say where a caller buffer lives (0 = host memory, 1 = HIP device memory of the
 *     model's device), mirroring dataHolder<T>{ptr, deviceType} (gbrl/src/cpp/types.h:252-270).
 *   - all compute runs on the GPU.  There is NO CPU fallback: step/predict fail with GBRL_HIP_E_NO_DEVICE
 *     when no HIP device is usable.  The `device` string of the reference ("cpu" / "cuda" / "gpu") only
 *     selects where predict() results are delivered by the Python binding.
 *   - a model is not thread-safe (like the reference object, binding.cpp:448); one model <-> one device.
 */
#ifndef GBRL_HIP_H
#define GBRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBRL_HIP_ABI_VERSION 1
#define GBRL_HIP_MAX_CHAR_SIZE 128 /* MAX_CHAR_SIZE, gbrl/src/cpp/types.h:56 */

typedef struct gbrl_hip_model gbrl_hip_model; /* opaque; replaces class GBRL (gbrl/src/cpp/gbrl.h) */

typedef enum {
    GBRL_HIP_OK = 0,
    GBRL_HIP_E_INVALID = -1,   /* bad argument / incompatible dimensions  (std::runtime_error in the reference) */
    GBRL_HIP_E_NO_DEVICE = -2, /* no usable HIP device: the product has no CPU path */
    GBRL_HIP_E_HIP = -3,       /* a HIP runtime call failed */
    GBRL_HIP_E_IO = -4,        /* file open / read / write error */
    GBRL_HIP_E_UNSUPPORTED = -5 /* valid in the reference, outside this build's scope (Adam, control variates, ...) */
} gbrl_hip_status;

/* enum values follow gbrl/src/cpp/types.h:110-181 so that the serialized metadata is byte-compatible */
enum { GBRL_HIP_SCORE_L2 = 0, GBRL_HIP_SCORE_COSINE = 1 };
enum { GBRL_HIP_GEN_UNIFORM = 0, GBRL_HIP_GEN_QUANTILE = 1 };
enum { GBRL_HIP_GROW_GREEDY = 0, GBRL_HIP_GROW_OBLIVIOUS = 1 };
enum { GBRL_HIP_ALGO_SGD = 0, GBRL_HIP_ALGO_ADAM = 1 };
enum { GBRL_HIP_SCHED_CONST = 0, GBRL_HIP_SCHED_LINEAR = 1 };

/* Constructor arguments: GBRL::GBRL(...) gbrl/src/cpp/gbrl.cpp:76-114, Python defaults binding.cpp:423-440 */
typedef struct {
    int32_t input_dim, output_dim, policy_dim, max_depth, min_data_in_leaf, n_bins, par_th;
    float cv_beta;
    int32_t split_score_func; /* GBRL_HIP_SCORE_*  */
    int32_t generator_type;   /* GBRL_HIP_GEN_*    */
    int32_t use_control_variates; /* must be 0: the reference's GPU path force-disables it too (gbrl.cpp:204-207) */
    int32_t batch_size;
    int32_t grow_policy;      /* GBRL_HIP_GROW_*   */
    int32_t verbose;
    int32_t device_ordinal;   /* HIP device to run on (the reference hard-codes 0, cuda_utils.cu:59); -1 = current */
    const char *learner_name;
} gbrl_hip_config;

/* ensembleMetaData, gbrl/src/cpp/types.h:218-242 -- exactly 80 bytes, written raw into .gbrl_model files */
typedef struct {
    int32_t n_leaves, n_trees, max_trees, max_leaves, max_trees_batch, max_leaves_batch;
    int32_t input_dim, output_dim, policy_dim, max_depth, min_data_in_leaf, n_bins, par_th;
    float cv_beta;
    int32_t verbose, batch_size;
    uint8_t use_cv, split_score_func, generator_type, grow_policy;
    int32_t n_num_features, n_cat_features, iteration;
} gbrl_hip_metadata;

/* optimizerConfig, gbrl/src/cpp/types.h:186-197 */
typedef struct {
    int32_t algo, scheduler;
    float init_lr, stop_lr;
    int32_t start_idx, stop_idx, T;
    float beta_1, beta_2, eps;
} gbrl_hip_optimizer;

/* ---- library / device --------------------------------------------------------------------------------- */
int gbrl_hip_abi_version(void);
/* number of usable HIP devices (0 when none); replaces GBRL::cuda_available (gbrl.cpp:542-548) */
int gbrl_hip_device_count(void);
const char *gbrl_hip_last_error(void);
/* device buffers handed to callers (predict results) -- replaces cudaMalloc/cudaFree in binding.cpp:208-219.
 * Freed buffers are recycled (a few buffers, <= 1 GiB); a recycled buffer is handed out only after a device
 * synchronisation, so a consumer kernel of its previous life cannot still be reading it. */
void *gbrl_hip_device_alloc(size_t bytes);              /* on the calling thread's current device */
void *gbrl_hip_device_alloc_on(int device, size_t bytes); /* on `device` (the buffer pool is keyed by device) */
void gbrl_hip_device_free(void *ptr);

/* ---- lifetime ----------------------------------------------------------------------------------------- */
gbrl_hip_model *gbrl_hip_create(const gbrl_hip_config *cfg);            /* GBRL::GBRL          gbrl.cpp:76-114   */
gbrl_hip_model *gbrl_hip_clone(const gbrl_hip_model *other);            /* GBRL::GBRL(GBRL&)   gbrl.cpp:125-148  */
gbrl_hip_model *gbrl_hip_load(const char *filename);                    /* GBRL::loadFromFile  gbrl.cpp:1175-1250 */
int gbrl_hip_save(gbrl_hip_model *m, const char *filename);             /* GBRL::saveToFile    gbrl.cpp:1130-1173 */
void gbrl_hip_destroy(gbrl_hip_model *m);                               /* GBRL::~GBRL         gbrl.cpp:150-165  */

/* ---- state the hot path reads ------------------------------------------------------------------------- */
int gbrl_hip_set_bias(gbrl_hip_model *m, const float *bias, int n, int on_device);               /* gbrl.cpp:213-240 */
int gbrl_hip_set_feature_weights(gbrl_hip_model *m, const float *w, int n, int on_device);       /* gbrl.cpp:242-269 */
int gbrl_hip_set_feature_mapping(gbrl_hip_model *m, const int32_t *feature_mapping,
                                 const uint8_t *mapping_numerics, int n);                        /* gbrl.cpp:271-316 */
int gbrl_hip_set_optimizer(gbrl_hip_model *m, const gbrl_hip_optimizer *opt);                    /* gbrl.cpp:452-525 */
int gbrl_hip_get_metadata(const gbrl_hip_model *m, gbrl_hip_metadata *out);                      /* binding.cpp:309-328 */
int gbrl_hip_get_bias(const gbrl_hip_model *m, float *out);                                      /* gbrl.cpp:318-330 */
int gbrl_hip_get_feature_weights(const gbrl_hip_model *m, float *out);                           /* gbrl.cpp:332-344 */
int gbrl_hip_get_feature_mapping(const gbrl_hip_model *m, int32_t *feature_mapping, uint8_t *mapping_numerics);
int gbrl_hip_num_optimizers(const gbrl_hip_model *m);
int gbrl_hip_get_optimizer(const gbrl_hip_model *m, int idx, gbrl_hip_optimizer *out);           /* binding.cpp:393-419 */
const char *gbrl_hip_learner_name(const gbrl_hip_model *m);
/* GBRL::get_ensemble_data (gbrl.cpp:1344-1356): copies of the ensembleData arrays (types.h:279-304).  Sizes:
 * T=n_trees, L=n_leaves, S = T (oblivious) or L (greedy), md=max_depth, D=output_dim, in=input_dim.  Any
 * pointer may be NULL. */
int gbrl_hip_get_ensemble(const gbrl_hip_model *m,
                          int32_t *tree_indices /*[T]*/, int32_t *depths /*[S]*/, float *values /*[L*D]*/,
                          int32_t *feature_indices /*[S*md]*/, float *feature_values /*[S*md]*/,
                          float *edge_weights /*[L*md]*/, uint8_t *is_numerics /*[S*md]*/,
                          uint8_t *inequality_directions /*[L*md]*/, char *categorical_values /*[S*md*128]*/,
                          int32_t *reverse_num_feature_mapping /*[in]*/, int32_t *reverse_cat_feature_mapping /*[in]*/);

/* ---- THE HOT PATH ------------------------------------------------------------------------------------- */
/* GBRL::step (gbrl.cpp:939-981) == Fitter::step_cpu semantics (fitter.cpp:50-115), computed on the GPU:
 * fits ONE tree to `grads` and appends it to the ensemble.
 *   obs      float32 [n_samples, n_num_features] row-major, or NULL when n_num_features == 0
 *   cat_obs  char    [n_samples, n_cat_features, 128] (NUL-padded strings), or NULL
 *   grads    float32 [n_samples, output_dim] row-major
 * Buffers are borrowed for the duration of the call only. */
int gbrl_hip_step(gbrl_hip_model *m, const float *obs, int obs_on_device, const char *cat_obs,
                  int cat_on_device, const float *grads, int grads_on_device, int n_samples,
                  int n_num_features, int n_cat_features);

/* GBRL::fit (gbrl.cpp:983-1104) == Fitter::fit_cpu semantics (fitter.cpp:117-261), MultiRMSE loss (loss.cpp:42-56), computed
 * on the GPU: bias = column means of `targets`; split candidates from the WHOLE data set once; then `iterations` boosting
 * rounds over consecutive batches of metadata.batch_size rows (predict over trees [0, i) -> gradients pred - target -> one
 * tree); *loss_out = sqrt(0.5 * sum (pred - target)^2 / n_samples) over the whole data set afterwards.  shuffle != 0 fits
 * a randomly permuted copy (seeded from std::random_device like the reference).  */
int gbrl_hip_fit(gbrl_hip_model *m, const float *obs, int obs_on_device, const char *cat_obs, int cat_on_device,
                 const float *targets, int targets_on_device, int n_samples, int n_num_features, int n_cat_features,
                 int iterations, int shuffle, float *loss_out);

/* GBRL::predict (gbrl.cpp:369-422) == Predictor::predict_cpu semantics (predictor.cpp:122-265) + SGDOptimizer::step
 * (optimizer.cpp:110-118): out[i,:] = bias - sum_t lr_k * leaf_value(i, t) over trees [start_tree, stop_tree)
 * (stop_tree == 0 means n_trees).  `out` is float32 [n_samples, output_dim], host or device per out_on_device. */
int gbrl_hip_predict(gbrl_hip_model *m, const float *obs, int obs_on_device, const char *cat_obs,
                     int cat_on_device, int n_samples, int n_num_features, int n_cat_features,
                     int start_tree, int stop_tree, float *out, int out_on_device);

/* Extension (no counterpart in the reference, which re-compares the 128-byte cells of every row inside every predict call,
 * predictor.cpp:231-265 / 188-229): a serving loop that predicts the SAME categorical batch repeatedly, or produces its categorical columns
 * from a small vocabulary, encodes them once.  gbrl_hip_encode_categorical writes int32 ids[n_samples, n_cat_features] (0 = a category no
 * condition of the model mentions) and a token identifying the model's category dictionary; gbrl_hip_predict_encoded is gbrl_hip_predict
 * with those ids in place of the cells and fails with GBRL_HIP_E_INVALID when the token is not the model's current one (a later tree
 * mentioned a new category, or the ids belong to another model): encode again.  Same results as gbrl_hip_predict, bit for bit. */
int gbrl_hip_encode_categorical(gbrl_hip_model *m, const char *cat_obs, int cat_on_device, int n_samples, int n_cat_features,
                                int32_t *ids_out, int ids_on_device, uint64_t *dictionary_token);
int gbrl_hip_predict_encoded(gbrl_hip_model *m, const float *obs, int obs_on_device, const int32_t *cat_ids, int ids_on_device,
                             uint64_t dictionary_token, int n_samples, int n_num_features, int n_cat_features,
                             int start_tree, int stop_tree, float *out, int out_on_device);

/* ---- row-sharded multi-GPU (new; the reference is single-GPU) ------------------------------------------ */
/* One process per GPU, each holding a contiguous block of rows.  When hooks are installed, step() calls them at
 * its exchange points so that every rank grows the identical tree; predict() needs no exchange.  Buffers are
 * DEVICE pointers; the hook must return only after the reduced result is visible on the HIP null stream
 * ordering used by the model.  sum hooks are exact (integers), so 1/2/4/8-GPU trees are
 * bit-identical.  Install NULL hooks to go back to single-GPU. */
typedef struct {
    void *ctx;
    int world_size, rank;
    int (*allreduce_sum_i64)(void *ctx, int64_t *dev_buf, size_t count);
    int (*allreduce_sum_f64)(void *ctx, double *dev_buf, size_t count);
    int (*allreduce_max_f32)(void *ctx, float *dev_buf, size_t count);
    int (*allreduce_min_f32)(void *ctx, float *dev_buf, size_t count);
} gbrl_hip_collective;
int gbrl_hip_set_collective(gbrl_hip_model *m, const gbrl_hip_collective *hooks);

/* Native exchange (preferred on multi-GPU nodes): the model creates its own RCCL communicator and enqueues every all-reduce
 * on its stream -- no host synchronisation at the exchange points.  RCCL is bound at run time (the librccl.so the process
 * already loaded, e.g. PyTorch's, else the system one).  Rank 0 calls gbrl_hip_rccl_unique_id and hands the 128 bytes to the
 * other ranks by any means (torch.distributed.broadcast in gbrl_amd/dist.py); then EVERY rank calls gbrl_hip_set_rccl
 * (a collective call: it returns when all ranks have joined).  Replaces hooks installed earlier; installing hooks later
 * destroys the communicator. */
int gbrl_hip_rccl_available(void);   /* 1 when an RCCL library could be bound in this process */
int gbrl_hip_rccl_unique_id(void *id128);
int gbrl_hip_set_rccl(gbrl_hip_model *m, const void *id128, int world_size, int rank);
/* The same with flags.  GBRL_HIP_RCCL_KEEP_WORLD1: keep the row-sharded code path (every exchange enqueued on the stream) at world size 1,
 * which gbrl_hip_set_rccl drops as "nothing to exchange" -- what bench.py's `collective` leg measures (the fixed cost of that path before a
 * byte crosses xGMI).  No reference counterpart (the reference has no collective). */
#define GBRL_HIP_RCCL_KEEP_WORLD1 1u
int gbrl_hip_set_rccl_flags(gbrl_hip_model *m, const void *id128, int world_size, int rank, unsigned flags);

/* ---- measurement -------------------------------------------------------------------------------------- */
/* ---- inspection, served from the host copy of the ensemble ---------------------- */
/* Linear TreeSHAP of one tree / of the whole ensemble: GBRL::tree_shap gbrl.cpp:1269-1303, GBRL::ensemble_shap gbrl.cpp:1305-1342,
 * algorithm shap.cpp:38-364.  Evaluated on the HIP device when one is usable (same bits as the host evaluation, which serves
 * machines without a GPU).  HOST pointers only (the reference's binding accepts NumPy arrays only, binding.cpp:985-1117).
 * obs [n_samples][n_num_features] f32, cat_obs [n_samples][n_cat_features][128] bytes (either may be NULL when the model has no
 * such features); norm_values [(max_depth+1)][max_depth], base_poly [max_depth], offset [max_depth][max_depth] as built by
 * gbrl/common/utils.py:317-371.  out [n_samples][n_num_features + n_cat_features][output_dim] is OVERWRITTEN
 * (left untouched by gbrl_hip_ensemble_shap when the ensemble has no trees: the feature counts are unknown until the first step). */
int gbrl_hip_tree_shap(const gbrl_hip_model *m, int tree_idx, const float *obs, const char *cat_obs, int n_samples,
                       const float *norm_values, const float *base_poly, const float *offset, float *out);
int gbrl_hip_ensemble_shap(const gbrl_hip_model *m, const float *obs, const char *cat_obs, int n_samples,
                           const float *norm_values, const float *base_poly, const float *offset, float *out);
/* GBRL::exportModel gbrl.cpp:1106-1128 (text: export_ensemble_data types.cpp:409-679).  export_format "float"|"fxp8"|"fxp16",
 * export_type "full"|"compact"; NULL strings mean the binding's defaults ("", "float", "full", "").  Oblivious models only. */
int gbrl_hip_export(const gbrl_hip_model *m, const char *filename, const char *modelname, const char *export_format,
                    const char *export_type, const char *prefix);
/* GBRL::print_tree gbrl.cpp:1357-1391 (tree_idx -1 = last tree) and GBRL::print_ensemble_metadata gbrl.cpp:1254-1267: the same
 * text, written to stdout.  device_name is what get_device() reports ("cpu" | "cuda"). */
int gbrl_hip_print_tree(const gbrl_hip_model *m, int tree_idx);
int gbrl_hip_print_ensemble_metadata(const gbrl_hip_model *m, const char *device_name);
/* GBRL::plot_tree gbrl.cpp:1409-1547 needs Graphviz, which this image does not have: always GBRL_HIP_E_UNSUPPORTED with the
 * message of the reference's own no-Graphviz build ("GBRL compiled without Graphviz! Cannot plot model"). */
int gbrl_hip_plot_tree(const gbrl_hip_model *m, int tree_idx, const char *filename);
/* what get_ensemble_data()["alloc_data_size"] shows in the reference: the size of the exact-size copy it hands out
 * (copy_ensemble_data types.cpp:322-384, binding.cpp:382) */
size_t gbrl_hip_alloc_data_size(const gbrl_hip_model *m);

/* Per-phase GPU time of the LAST step()/predict() call, measured with HIP events on the model's stream.
 * names/ms hold up to `cap` entries; returns the number of phases. */
int gbrl_hip_last_phase_times(const gbrl_hip_model *m, const char **names, float *ms, int cap);
/* level 0 (default): no events.  1: only the dominant kernel's launches (histogram build in step(), the traversal kernel in
 * predict()) are timed -- step(): the dispatch's own begin/end timestamps (hipExtLaunchKernelGGL events), free for a timed region.  2: every phase is bracketed
 * (diagnostic: each record costs a few microseconds of stream bubble, ~60 records per step). */
int gbrl_hip_set_profiling(gbrl_hip_model *m, int level);
/* Diagnostics (tests): the near-tie replay kernel (csrc/neartie.hip) on ONE node given explicitly -- the float32 score the reference's
 * TreeNode::splitScoreCosine / splitScoreL2 (node.cpp:187-251, 321-376) gives the split {rows with goes_right} | {the others} of the rows
 * with in_node set, and the node's parent score (scoreCosine / scoreL2, split_candidate_generator.cpp:262-320).  Host pointers: grads
 * [n_rows][output_dim]; in_node, goes_right [n_rows] bytes; meanden nullable [2 * output_dim] (L2: column means | std + 1e-8f, the
 * standardisation applied to grads before scoring); out_scores[2] = split score | parent score.  n_rows <= 65536. */
int gbrl_hip_replay_scores(const float *grads, const uint8_t *in_node, const uint8_t *goes_right, int n_rows, int output_dim,
                           const float *meanden, int cosine, int min_data_in_leaf, float *out_scores);

/* Diagnostics for the tests: sequential float32 sums -- s = start; for every element in order: s = (float)(s + x) -- of `n_chains` arrays stored
 * back to back in `x` (`lens[i]` elements each), evaluated by the parallel block-summary kernels of seqsum.hip; the results must equal the plain
 * loop bit for bit.  Host pointers; starts nullable (zeros); n_slow_blocks nullable: how many 256-element blocks took the serial fallback.
 * No reference counterpart (the reference IS the plain loop: node.cpp:336-352). */
int gbrl_hip_seq_sums(const float *x, const uint32_t *lens, const float *starts, int n_chains, float *out, uint32_t *n_slow_blocks);

/* ---- device / stream contract (new; the reference pins everything to device 0 and the null stream, cuda_types.cu:32-106) -- */
/* The device the model computes on: the ordinal given at creation, or -- for -1 -- the calling thread's current device at the
 * first call that needs it (it is latched by this call too).  -1 when no HIP device is usable.  A caller that hands out
 * predict results as device buffers must allocate them on THIS device (gbrl_hip_device_alloc_on) and label them with it. */
int gbrl_hip_device_ordinal(gbrl_hip_model *m);
/* Stream ordering.  By default the model owns a BLOCKING stream: its work is ordered with the legacy null stream only, so a
 * caller that runs on a non-default (or per-thread default) stream must either synchronise that stream around step()/predict()
 * or hand it over here: with a non-null `hip_stream` (a hipStream_t of the model's device) every kernel, copy and collective
 * of this model is enqueued on it -- inputs produced on that stream and outputs consumed on it need no further
 * synchronisation.  step()/predict() still return after their own work has completed (they read results back).  NULL
 * restores the model's own stream. */
int gbrl_hip_set_stream(gbrl_hip_model *m, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* GBRL_HIP_H */
