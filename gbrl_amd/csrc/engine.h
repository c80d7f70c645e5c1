// engine.h -- device engine: owns the host model, its device mirror, the per-step workspace and the HIP stream, and
// drives the kernels of kernels.hip for GBRL::step / GBRL::predict.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <array>
#include <chrono>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "model.h"

namespace gbrl {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct NoDeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct InvalidArgument : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

void hip_check(hipError_t e, const char *what);

// grow-only device buffer
class DevBuf {
   public:
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf();
    void *ensure(size_t bytes);  // contents are NOT preserved when it grows
    void *ensure_keep(size_t bytes, size_t keep_bytes, hipStream_t s);  // preserves the first keep_bytes
    template <typename T>
    T *as() const { return static_cast<T *>(ptr_); }
    const void *raw() const { return ptr_; }
    size_t capacity() const { return cap_; }
    void release();

   private:
    void *ptr_ = nullptr;
    size_t cap_ = 0;
};

// grow-only pinned host buffer (async copies from/to it do not stage through pageable memory)
class PinnedBuf {
   public:
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf();
    void *ensure(size_t bytes);

   private:
    void *ptr_ = nullptr;
    size_t cap_ = 0;
};

namespace detail {
struct GrowCtx;
struct HNode;
// a categorical split candidate: class `cls` of feature `feat` is the category `name` (the raw 128-byte cell, types.h:55-58)
struct CatCandidate {
    int feat;
    std::array<char, 128> name;
    int cls;
    CatCandidate(int f, const char *cell, int c) : feat(f), cls(c) { std::memcpy(name.data(), cell, 128); }
};
// A distinct (categorical feature, cell) pair the engine has met in some step: the hash of its key in the reference's candidate
// container (std::hash of cell + "_" + feature, split_candidate_generator.cpp:121) is computed once.
struct CatItem { int feat; int next; uint64_t lhash; size_t std_hash; char name[128]; };
}  // namespace detail

class Engine {
   public:
    explicit Engine(const gbrl_hip_config &cfg);
    explicit Engine(Model &&loaded, int device_ordinal);
    Engine(const Engine &other);  // deep copy of the model (GBRL::GBRL(GBRL&)); fresh device state
    ~Engine();

    Model model;

    void step(const float *obs, bool obs_dev, const char *cat, bool cat_dev, const float *grads, bool grads_dev, int n,
              int n_num, int n_cat);
    void predict(const float *obs, bool obs_dev, const char *cat, bool cat_dev, int n, int n_num, int n_cat, int start_tree,
                 int stop_tree, float *out, bool out_dev);
    // Extension (not in the reference, which compares the 128-byte cells inside every predict call): the dictionary ids of a batch of
    // categorical cells, and predict from such ids.  `token` identifies the model's category dictionary (a hash of its entries in order): ids are
    // valid for any model whose dictionary is the same (this model until a later tree mentions a new category, its clones, its saved file).
    void encode_categorical(const char *cat, bool cat_dev, int n, int n_cat, int32_t *ids_out, bool out_dev, uint64_t *token);
    void predict_encoded(const float *obs, bool obs_dev, const int32_t *cat_ids, bool ids_dev, uint64_t token, int n, int n_num, int n_cat,
                         int start_tree, int stop_tree, float *out, bool out_dev);
    // GBRL::fit (gbrl.cpp:983-1104) + Fitter::fit_cpu (fitter.cpp:117-261): bias = mean(targets), split candidates from the
    // WHOLE data set once, then `iterations` boosting rounds over consecutive batches of metadata.batch_size rows
    // (predict -> MultiRMSE gradients -> one tree per batch); returns the final MultiRMSE loss on the whole data set.
    float fit(const float *obs, bool obs_dev, const char *cat, bool cat_dev, const float *targets, bool targets_dev, int n, int n_num,
              int n_cat, int iterations, bool shuffle);

    // Linear TreeSHAP of tree `tree_idx` (-1: every tree, summed in tree order) on the device; host pointers in and out, `out`
    // [n][n_num + n_cat][D] is overwritten.  Returns false when it was NOT computed (no HIP device, max_depth / output_dim beyond
    // the kernel's LDS budget, GBRL_HIP_SHAP_HOST=1): the caller then evaluates on the host (explain.cpp), which gives the same bits.
    bool shap_on_device(int tree_idx, const float *obs, const char *cat, int n, const float *norm, const float *base_poly,
                        const float *offset, float *out);

    void set_collective(const gbrl_hip_collective *hooks);
    // Native exchange: an RCCL communicator of this engine's own, collectives enqueued on its stream (no host sync).
    // id128 = gbrl_hip_rccl_unique_id() of rank 0, distributed by the caller.  Collective call (all ranks).
    void set_rccl(const void *id128, int world_size, int rank, bool keep_world1 = false);
    int device_ordinal();                 // latches the device like the first step()/predict() would
    void set_stream(hipStream_t s);       // nullptr: back to the engine's own blocking stream
    void set_profiling(int level) { profiling_ = level; }   // 0 off, 1 histogram build only (one launch in seven, every level in turn), 2 every phase
    void set_force_bisection(bool on) { force_bisection_ = on; }   // test hook: exercise the slow exact quantile path
    bool last_quantile_fallback() const { return last_quantile_fallback_; }
    const std::vector<std::pair<std::string, float>> &phase_times() const { return phases_; }

   private:
    void ensure_device();
    void sync_model_to_device();
    int32_t *encode_categorical_batch(const char *cat, bool cat_dev, int n, int n_cat);
    void predict_core(const float *obs, bool obs_dev, const char *cat, bool cat_dev, const int32_t *cat_ids, bool ids_dev, const uint64_t *token,
                      int n, int n_num, int n_cat, int start_tree, int stop_tree, float *out, bool out_dev);
    uint64_t cat_dict_token();
    void grow_tree(const detail::GrowCtx &c, std::vector<detail::HNode> &nodes, std::vector<int> &frontier, std::vector<int64_t> &acc,
                   double &leaf_scale);
    bool device_categorical_candidates(const char *dcells, const char *hcells, int N, int Fc, int B,
                                       std::vector<detail::CatCandidate> &cat_cands, std::vector<int> &cat_classes, bool launch_only = false);
    void sharded_categorical_ranking(const char *hcat, const float *hgrads, int N, int Fc, int D, int B, std::vector<detail::CatCandidate> &cat_cands,
                                     std::vector<uint16_t> &h_catcodes, std::vector<int> &cat_classes);
    void numeric_thresholds(const float *dobs, int N, int F, int B, long long n_global, const uint32_t *d_kt, float *d_thr,
                            uint32_t *d_thrkeys, int pass1_chunks = 0, uint16_t *d_codes_out = nullptr, bool *codes_written = nullptr);
    int64_t *quantile_cum_device(const std::vector<int64_t> &cum, long long n_global, int B);   // device copy of the quantile target ranks (cached)
    void phase_begin(bool key = false);
    void phase_end(const char *name, bool key = false);
    void phases_resolve();
    std::pair<hipEvent_t, hipEvent_t> kernel_events(const char *name, bool key);   // event pair for one dispatch's own timestamps

    int device_ordinal_ = -1;
    bool device_ready_ = false;
    hipStream_t stream_ = nullptr;        // the stream everything is enqueued on: own_stream_ or the caller's (set_stream)
    hipStream_t own_stream_ = nullptr;
    gbrl_hip_collective coll_{};
    bool has_coll_ = false;
    void *rccl_comm_ = nullptr;          // non-null: the exchanges below are RCCL calls on stream_
    enum class Red { SumI64, SumF64, MaxF32, MinF32 };
    void exchange(Red op, void *dev_buf, size_t count);   // all-reduce in place; stream-ordered (RCCL) or host-synchronous (hooks)
    // recv[0 .. count) = sum over ranks of their send[rank * count .. (rank + 1) * count)  (int64).  RCCL: ncclReduceScatter on the
    // stream (half the bytes of an all-reduce over the xGMI ring); hooks: all-reduce of the whole send buffer, then the own slice.
    void reduce_scatter_i64(int64_t *send, int64_t *recv, size_t count);
    size_t exch_bytes_ = 0, exch_calls_ = 0;   // bytes this rank hands to the transport per step() (diagnostic, reported as phases)
    static int radix_exchange_trampoline(void *self, int64_t *dev_buf, size_t count);

    // measurement
    int profiling_ = 0;
    unsigned key_calls_ = 0;   // key-kernel launches seen at profiling level 1 (every seventh carries an event pair)
    bool force_bisection_ = false, force_sample_select_ = false, force_host_categorical_ = false, force_radix_ = false, last_quantile_fallback_ = false;
    // fit(): numeric thresholds computed once from the whole data set and reused by every batch's step()
    std::vector<float> fixed_thr_;
    std::vector<detail::CatCandidate> fixed_cat_cands_;
    std::vector<int> fixed_cat_classes_;
    bool fixed_cat_valid_ = false;
    bool candidates_only_ = false;
    bool in_fit_ = false;   // predict() called from fit(): keep one accumulation chain per row in tree order (no tree-range split)
    std::vector<std::pair<std::string, float>> phases_;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool_;
    bool cat_launched_ = false;       // device_categorical_candidates(launch_only) has enqueued the first round of the scan
    hipEvent_t ev_level_ = nullptr;   // marks the per-level result read-back (GBRL_HIP_EVENT_RESULTS=1: the copy-engine path)
    // Per-step constants of numeric-only steps (feature slots, candidate weights / reference order / slot lookup): they depend on
    // (F, n_bins, growth policy, feature weights, feature mapping) only, so they are built and uploaded once and reused while those
    // stay the same and the device staging block is still the one they were uploaded to.
    struct StepConstCache {
        bool valid = false;
        int F = 0, B = 0, oblivious = 0;
        std::vector<float> fw;
        std::vector<int32_t> rev;
        std::vector<kern::FeatureSlot> slots;
        std::vector<int32_t> cand_ref, cand_slot;
        std::vector<float> cand_w;
        std::vector<int> ref_to_internal;
        const void *dev_base = nullptr;   // device staging block that holds the uploaded copy (nullptr: not uploaded yet)
        size_t stage_bytes = 0;
    } step_const_;
    DevBuf d_rows_iota_;              // 0, 1, 2, ...: the root's row list, generated when it grows and read-only afterwards
    const void *iota_ptr_ = nullptr;
    int iota_n_ = 0;
    DevBuf d_pub_done_;               // block counter of k_resolve_splits' in-kernel publication (zero between launches)
    const void *pub_done_ptr_ = nullptr;
    const void *leafacc_clean_ptr_ = nullptr;   // leaf accumulators known to be zero (handed back clean by the last publication)
    size_t leafacc_clean_bytes_ = 0;
    DevBuf d_codes_fm_;               // feature-major copy of the numeric class codes (kern::small_prep -> kern::small_grow)
    DevBuf d_am_s_, d_near_list_, d_near_ent_, d_near_rep_, d_near_nr_, d_near_maps_, d_near_pos_, d_near_nrb_, d_near_vals_, d_near_means_, d_near_sums_, d_near_chains_, d_near_rowsort_, d_near_tiles_;   // near-tie replay (kern::near_tie_replay): runner-up per arg-max block, candidate lists, ordered row lists, replayed scores
    bool force_level_loop_ = false;   // grow_tree: the one-launch kernel met a near-tie and handed the tree to the level loop
    bool small_grow_off_ = false;     // latched after a failed launch / an abandoned grid barrier of the one-launch kernel: this engine keeps to the level loop
    long long small_grow_fallbacks_ = 0;   // trees the level loop grew after such a failure (diagnostics)
    long long near_replays_ = 0, near_bailouts_ = 0, near_in_kernel_ = 0;   // levels replayed / one-launch trees handed over (diagnostics: phases at profiling level 2)
    DevBuf d_sg_bests_, d_sg_sync_, d_sg_near_;   // one-launch growth of RL-sized steps (kern::small_grow): per-level bests of every block, barrier words
    const void *sg_sync_ptr_ = nullptr;
    std::chrono::steady_clock::time_point prof_marks_[4]{};
    std::chrono::steady_clock::time_point prof_step_entry_{};   // measurement (GBRL_HIP_SMALL_GROW_PROF)
    int prep_launches_ = 0;           // diagnostic of the last step(): 1 = kern::small_prep ran, 3 = the separate preparation launches
    uint32_t level_seq_ = 0;          // sequence number of the last published level result block (0 is never published)
    std::vector<const char *> ev_names_;
    size_t ev_used_ = 0;

    DevBuf d_pred_partial_;
    DevBuf d_pred_slots_;     // leaf slots of the two-launch chain path (kern::predict_chain)
    DevBuf d_shap_ops_, d_shap_nodes_, d_shap_values_, d_shap_poly_, d_shap_out_;
    int shap_n_ops_ = 0;
    uint64_t shap_prog_version_ = ~0ull;

    // ---- per-step workspace (grow-only, reused across steps) ----
    DevBuf d_obs_, d_grads_, d_qg_, d_stat_, d_partials_f64_, d_meanden_, d_maxbits_;
    DevBuf d_thr_, d_thrkeys_, d_prefix_, d_trial_, d_counts_, d_cum_, d_minmax_;
    DevBuf d_selcnt_, d_kcls_, d_kt_, d_qflags_, d_splitters_, d_ccounts_, d_c2l_, d_tgt_list_, d_tgt_rank_, d_list_off_, d_qlists_;
    DevBuf d_radix_state_, d_radix_partial_, d_radix_global_, d_scales_;
    DevBuf d_codes_, d_catcodes_, d_rows_[2], d_chunks_, d_chunk_begin_;
    DevBuf d_hist_prev_, d_slotmap_, d_am_v_, d_am_i_, d_stage_const_, d_stage_a_, d_stage_b_, d_results_;
    PinnedBuf pin_const_, pin_a_, pin_b_, pin_res_, pin_thr_, pin_acc_, pin_cat_, pin_cat_dict_;
    DevBuf d_hist_partials_, d_hist_, d_hist_local_, d_hist_recv_, d_gather_, d_slots_, d_scores_, d_parent_, d_cand_w_, d_cand_ref_;
    DevBuf d_path_len_, d_path_slot_, d_path_val_, d_path_bin_, d_isroot_;
    DevBuf d_best_idx_, d_best_score_, d_splits_, d_ntotal_, d_nright_, d_cursors_, d_leafacc_, d_plan_, d_res_all_;
    PinnedBuf pin_res_all_, pin_cum_;
    long long cum_cache_n_ = -1;   // (global rows, n_bins) the device copy of the quantile target ranks was built for
    int cum_cache_b_ = -1;
    // ---- predict workspace + device mirror of the ensemble ----
    DevBuf d_pobs_, d_pcat_, d_pout_;
    DevBuf m_tree_indices_, m_depths_, m_feature_indices_, m_feature_values_, m_values_, m_is_numerics_, m_ineq_,
        m_cat_ids_, m_bias_, m_opt_start_, m_opt_stop_, m_opt_lr_, m_cond_pack_, m_grd_nodes_, m_grd_off_, m_values_sw_, m_cond_ra_;
    size_t up_trees_ = 0, up_leaves_ = 0, up_splits_ = 0;  // how much of the append-only arrays is already on the device
    uint64_t mirror_version_ = ~0ull;
    // dictionary of the categorical strings that occur in the model's conditions: (cat feature, string) -> id >= 1
    std::vector<int32_t> cat_ids_host_, cond_pack_host_, grd_nodes_host_, grd_off_host_, cond_ra_host_;
    std::vector<float> values_sw_host_;   // second-generation oblivious predict: see kern::PredictModel::values_sw
    // packed-code predict (kern::predict_pc): the ensemble's code book -- per numeric feature the sorted distinct thresholds, per
    // mentioned category a bit slot, per tree level a (word, shift, T) record -- rebuilt when the model has changed
    bool ensure_pc_book(int n_num, int n_cat);
    DevBuf m_pc_cond_, m_pc_thr_, m_pc_thr_off_, m_pc_cat_slot_, m_pc_word_cols_, d_pc_rows_;
    size_t pc_version_ = static_cast<size_t>(-1);
    int pc_f_ = -1, pc_fc_ = -1, pc_wn_ = 0, pc_nw_ = 0, pc_row_words_ = 0, pc_iters_ = 1;
    bool pc_ok_ = false;
    bool grd_ok_ = true;
    int grd_max_nodes_ = 0, grd_max_leaves_ = 1;
    size_t grd_up_nodes_ = 0;
    std::vector<std::pair<int, std::string>> cat_dict_;
    uint64_t dict_token_ = 0; size_t dict_token_size_ = static_cast<size_t>(-1);   // cat_dict_token(): cached per dictionary size (entries are only appended)
    DevBuf d_pcat_in_;   // pre-encoded ids handed over in host memory
    size_t dict_version_ = static_cast<size_t>(-1);   // cat_dict_.size() the device dictionary was built from
    int dict_fc_ = -1;
    DevBuf d_dict_off_, d_dict_hash_, d_dict_id_, d_dict_words_, d_pcells_;
    PinnedBuf pin_model_stage_;           // the slices sync_model_to_device appends, staged for one kern::stage_copy launch
    hipEvent_t ev_model_stage_ = nullptr; // behind that launch: the block is not refilled before it has been read
    DevBuf d_root_le_;
    const uint32_t *root_le_ = nullptr;   // this step's #{keys <= threshold} table (radix selection, one GPU), null otherwise
    DevBuf d_cat_keys_, d_cat_first_, d_cat_meta_, d_cat_lslot_, d_sdict_, d_cat_xchg_, d_cat_slotq_, d_cat_clsq_;
    PinnedBuf pin_cat_cls_;
    uint32_t cat_pub_seq_ = 0;                          // sequence word of k_cat_publish's completion flag
    // ordinary steps on one GPU: the class codes come from the scan's own tables (k_cat_step_codes_table) instead of a dictionary
    struct { bool valid = false; const uint64_t *keys = nullptr; const int32_t *slot_q = nullptr, *cls_of_q = nullptr; int log2_cap = 0; } cat_table_;
    // the step's candidate dictionary inside d_sdict_ (one upload): per categorical feature the entries sorted by raw hash
    const int32_t *sdict_off_ = nullptr, *sdict_cls_ = nullptr;
    const uint64_t *sdict_hash_ = nullptr, *sdict_words_ = nullptr;
    int cat_log2_hint_ = 20;                            // log2 of the per-feature table size the next step starts with (20: the full size)
    int cat_publish_guess_ = 256;                       // records the next step publishes with its header (the last count + 25 %)
    void verify_pending_categories();
    std::vector<std::pair<int, const char *>> cat_pending_;   // (item, published cell) pairs whose bytes are still to be compared this step
    bool cat_clash_ = false;
    long long cat_clash_redos_ = 0;   // steps grown a second time on the host scan after a 64-bit hash clash (diagnostics)
    std::vector<detail::CatItem> cat_items_;            // every distinct (feature, cell) met so far
    std::vector<uint64_t> cat_tab_key_;                 // (raw hash, feature) -> head of the chain through CatItem::next: open-addressed
    std::vector<int32_t> cat_tab_id_;                   //   table (keys | item ids, -1 = empty), at most half full
    std::vector<char> cat_host_;                        // the published distinct-cell block, copied out of the pinned mapping
    std::vector<uint32_t> cat_seen_;                    // per item: tag of the last replay that inserted it
    uint32_t cat_seen_tag_ = 0;
    DevBuf d_fit_cells_, d_fit_cells2_;
    DevBuf d_fit_obs_, d_fit_targets_, d_fit_obs2_, d_fit_targets2_, d_fit_perm_, d_fit_preds_, d_fit_grads_, d_fit_zero_;
};

}  // namespace gbrl
