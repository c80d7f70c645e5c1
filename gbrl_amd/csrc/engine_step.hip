// engine_step.hip -- host orchestration of the HIP kernels for GBRL::step and GBRL::fit (see engine.h).
//
// step() restates Fitter::step_cpu (gbrl/src/cpp/fitter.cpp:50-115) as a histogram algorithm:
//   1. gradient statistics + fixed-point quantisation of the build gradients           (A2)
//   2. split candidates: uniform (min/max) or quantile (exact order statistics)          (A3, A4); categorical on the host (A5)
//   3. observations -> per-feature class codes (once per step)
//   4. level-synchronous growth: for every frontier node build (count, sum g[D]) per (feature, class) in LDS, reduce to
//      exact int64 histograms, score every candidate from suffix sums, pick the split, partition the row list   (A6-A10)
//   5. leaf values = exact mean of the raw gradients per leaf                              (A11)
// The reference grows greedy trees depth-first; the split chosen for a node depends only on that node's rows, so growing
// level by level and emitting the leaves in depth-first (left first) order afterwards gives the identical tree.
#include "engine_step_detail.h"

namespace gbrl {

namespace detail {

// ---- A5: categorical candidates on the host, exactly as processCategoricalCandidates (split_candidate_generator.cpp:117-163):
// same container, same insertion order => same candidate order (Q8).  cat_classes[f] = number of candidate categories of
// feature f (class ids 1..), h_catcodes[i*Fc+f] = class of the cell (0: not a candidate).
void categorical_candidates(const char *hcat, const float *hgrads, int N, int Fc, int D, int B, std::vector<CatCandidate> &cat_cands,
                            std::vector<uint16_t> &h_catcodes, std::vector<int> &cat_classes) {
    struct Info { float total = 0.f; int count = 0; int feat = 0; std::string name; };
    std::vector<float> norms(N, 0.0f);
    for (int i = 0; i < N; ++i) {  // calculate_squared_norm (math_ops.cpp:726-749), contracted like the reference build
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) { const float g = hgrads[static_cast<size_t>(i) * D + d]; acc = fmaf(g, g, acc); }
        norms[i] = acc;
    }
    std::unordered_map<std::string, Info> uniq;
    for (int f = 0; f < Fc; ++f)
        for (int i = 0; i < N; ++i) {
            std::string name(hcat + (static_cast<size_t>(i) * Fc + f) * kCat, kCat);
            Info &ci = uniq[name + "_" + std::to_string(f)];
            ci.total += norms[i];
            ci.count += 1;
            ci.feat = f;
            ci.name = name;
        }
    std::vector<std::pair<std::string, float>> vec;
    for (const auto &kv : uniq) vec.emplace_back(kv.first, kv.second.total / kv.second.count);
    int n_unique = static_cast<int>(vec.size());
    if (n_unique > Fc * B) {
        std::sort(vec.begin(), vec.end(), [](const std::pair<std::string, float> &a, const std::pair<std::string, float> &b) {
            return a.second > b.second;
        });
        n_unique = Fc * B;
    }
    std::unordered_map<std::string, int> cls_of;  // key -> class id within its feature
    for (int i = 0; i < n_unique; ++i) {
        const Info &ci = uniq[vec[i].first];
        const int cls = ++cat_classes[ci.feat];
        if (cls > 65534) throw Unsupported("more than 65534 candidate categories in one feature");
        cat_cands.emplace_back(ci.feat, ci.name.data(), cls);
        cls_of[vec[i].first] = cls;
    }
    h_catcodes.assign(static_cast<size_t>(N) * Fc, 0);
    for (int i = 0; i < N; ++i)
        for (int f = 0; f < Fc; ++f) {
            std::string key(hcat + (static_cast<size_t>(i) * Fc + f) * kCat, kCat);
            key += "_" + std::to_string(f);
            auto it = cls_of.find(key);
            if (it != cls_of.end()) h_catcodes[static_cast<size_t>(i) * Fc + f] = static_cast<uint16_t>(it->second);
        }
}

// ---- A10/A11: the grown tree joins the ensemble (update_ensemble_per_leaf / per_tree, fitter.cpp:493-542) with exact leaf
// means of the raw gradients (acc[node] = int64 fixed-point sums | count; fitter.cpp:545-582).
void append_tree(Model &model, const std::vector<HNode> &nodes, const std::vector<int> &frontier, const std::vector<int64_t> &acc,
                 double leaf_scale, const std::vector<CatCandidate> &cat_cands) {
    gbrl_hip_metadata &md = model.meta;
    const bool oblivious = model.oblivious();
    const int MD = md.max_depth, D = md.output_dim;
// leaf order: oblivious = level order of the last level (child slots 2k, 2k+1, fitter.cpp:469-470); greedy = depth-first,
// left first (fitter.cpp:364-365)
std::vector<int> leaf_order;
if (oblivious) {
    if (nodes.size() == 1) leaf_order.push_back(0);
    else leaf_order = frontier;
} else {
    std::vector<int> stack{0};
    while (!stack.empty()) {
        const int id = stack.back();
        stack.pop_back();
        if (nodes[id].left < 0) { leaf_order.push_back(id); continue; }
        stack.push_back(nodes[id].right);
        stack.push_back(nodes[id].left);
    }
}

// ---- append to the ensemble (update_ensemble_per_leaf / per_tree, fitter.cpp:493-542) -----------------------------
model.begin_tree();
const size_t tree = md.n_trees;
model.tree_indices.push_back(md.n_leaves);
auto write_conditions = [&](const HNode &nd, size_t split_row, size_t leaf_row) {
    for (size_t i = 0; i < nd.path.size(); ++i) {
        const HCond &c = nd.path[i];
        if (c.is_cat && c.cat_cand >= 0)
            std::memcpy(&model.categorical_values[(split_row * MD + i) * kCat], cat_cands[c.cat_cand].name.data(), kCat);
        model.is_numerics[split_row * MD + i] = c.is_cat ? 0 : 1;
        model.feature_indices[split_row * MD + i] = c.feat_idx;
        model.feature_values[split_row * MD + i] = c.value;
        model.inequality_directions[leaf_row * MD + i] = c.dir ? 1 : 0;
        model.edge_weights[leaf_row * MD + i] = c.edge_w;
    }
};
const size_t n_new = leaf_order.size();
const size_t L0 = md.n_leaves;
const size_t S_new = oblivious ? tree + 1 : L0 + n_new;
model.depths.resize(S_new, 0);
model.feature_indices.resize(S_new * MD, 0);
model.feature_values.resize(S_new * MD, 0.0f);
model.is_numerics.resize(S_new * MD, 0);
model.categorical_values.resize(S_new * MD * kCat, 0);
model.values.resize((L0 + n_new) * D, 0.0f);
model.edge_weights.resize((L0 + n_new) * MD, 0.0f);
model.inequality_directions.resize((L0 + n_new) * MD, 0);
for (size_t q = 0; q < n_new; ++q) {
    const HNode &nd = nodes[leaf_order[q]];
    const size_t leaf_row = L0 + q;
    if (oblivious) {
        model.depths[tree] = nd.depth;
        write_conditions(nd, tree, leaf_row);
    } else {
        model.depths[leaf_row] = nd.depth;
        write_conditions(nd, leaf_row, leaf_row);
    }
    const int64_t *a = &acc[static_cast<size_t>(leaf_order[q]) * (D + 1)];
    const int64_t cnt = a[D];
    for (int d = 0; d < D; ++d) {
        float v = 0.0f;
        if (cnt > 0 && nd.depth > 0)  // fitter.cpp:574-578; depth-0 leaf keeps 0 (Q7)
            v = static_cast<float>((static_cast<double>(a[d]) / leaf_scale) / static_cast<double>(cnt));
        model.values[leaf_row * D + d] = v;
    }
}
md.n_leaves += static_cast<int32_t>(n_new);
md.n_trees += 1;
md.iteration += 1;  // fitter.cpp:114
++model.version;
}

}  // namespace detail

// ======================================================================================================== step
void Engine::step(const float *obs, bool obs_dev, const char *cat, bool cat_dev, const float *grads, bool grads_dev, int n,
                  int n_num, int n_cat) {
    gbrl_hip_metadata &md = model.meta;
    // GBRL::step, gbrl.cpp:946-958
    if (md.iteration == 0) { md.n_num_features = n_num; md.n_cat_features = n_cat; }
    if (n_num != md.n_num_features || n_cat != md.n_cat_features) throw InvalidArgument("Incompatible dataset");
    if (n_num + n_cat != md.input_dim) throw InvalidArgument("Total number of features != correct input dim");
    if (n <= 0 || grads == nullptr) throw InvalidArgument("Cannot call step without grads!");
    if (n_num > 0 && obs == nullptr) throw InvalidArgument("Cannot call step without obs!");
    if (n_cat > 0 && cat == nullptr) throw InvalidArgument("Cannot call step without cat_obs!");
    if (md.max_depth > kern::kMaxPath) throw Unsupported("max_depth > 32 is not supported");
    if (md.n_bins < 1 || md.n_bins > 65534) throw Unsupported("n_bins must be in [1, 65534]");
    ensure_device();
    prof_step_entry_ = std::chrono::steady_clock::now();
    ev_used_ = 0;
    ev_names_.clear();
    exch_bytes_ = 0;
    exch_calls_ = 0;
    if (const char *e = hooks::raw(hooks::FORCE_BISECTION)) force_bisection_ = e[0] == '1';   // test hook
    if (const char *e = hooks::raw(hooks::HOST_CATEGORICAL)) force_host_categorical_ = e[0] == '1';   // test hook: host scan of every cell
    if (const char *e = hooks::raw(hooks::QUANTILE_RADIX)) force_radix_ = e[0] == '1';   // test hook: radix multi-select also for small batches
    if (const char *e = hooks::raw(hooks::QUANTILE_SAMPLE)) force_sample_select_ = e[0] == '1';   // test hook: the sample/splitter selection on one GPU
    hipStream_t s = stream_;
    const int N = n, F = n_num, Fc = n_cat, D = md.output_dim, B = md.n_bins, MD = md.max_depth;
    const bool cosine = md.split_score_func == GBRL_HIP_SCORE_COSINE;
    const bool oblivious = model.oblivious();
    const int world = has_coll_ ? coll_.world_size : 1;

    // global row count (rows are sharded over ranks): it travels with the first gradient-statistics message below (round 6: no exchange and no
    // idle device for it); until then n_global holds this rank's count and nothing reads it
    long long n_global = N;

    // ---- inputs on the device -------------------------------------------------------------------------------------
    phase_begin();
    const float *dobs = obs;
    if (F > 0 && !obs_dev) {
        dobs = static_cast<float *>(d_obs_.ensure(sizeof(float) * N * F));
        hip_check(hipMemcpyAsync(const_cast<float *>(dobs), obs, sizeof(float) * N * F, hipMemcpyHostToDevice, s), "H2D obs");
    }
    const float *dgrads = grads;
    if (!grads_dev) {
        dgrads = static_cast<float *>(d_grads_.ensure(sizeof(float) * N * D));
        hip_check(hipMemcpyAsync(const_cast<float *>(dgrads), grads, sizeof(float) * N * D, hipMemcpyHostToDevice, s), "H2D grads");
    }
    // categorical cells on the device: the distinct categories of the batch are found there (device_categorical_candidates);
    // the host-side scan of every cell is only the fallback
    const char *dcells = cat;
    if (Fc > 0 && !cat_dev) {
        char *t = static_cast<char *>(d_pcells_.ensure(static_cast<size_t>(N) * Fc * kCat));
        hip_check(hipMemcpyAsync(t, cat, static_cast<size_t>(N) * Fc * kCat, hipMemcpyHostToDevice, s), "H2D cat cells");
        dcells = t;
    }
    phase_end("inputs");
    prof_marks_[0] = std::chrono::steady_clock::now();
    // the scan for the batch's distinct categorical cells goes first: its result is read by the HOST (device_categorical_candidates)
    const bool cat_early = Fc > 0 && !fixed_cat_valid_ && (has_coll_ || !force_host_categorical_);
    cat_launched_ = false;   // (a step that threw between the two stages must not be resumed)
    if (cat_early) {
        std::vector<CatCandidate> none;
        std::vector<int> none_classes(Fc, 0);
        (void)device_categorical_candidates(dcells, cat_dev ? nullptr : cat, N, Fc, B, none, none_classes, /*launch_only=*/true);
    }

    // ---- 1. gradient statistics and quantisation (A2) -----------------------------------------------------------------
    phase_begin();
    const size_t n_el = static_cast<size_t>(N) * D;
    float *d_meanden = static_cast<float *>(d_meanden_.ensure(sizeof(float) * 2 * D));
    const float *d_mean = nullptr, *d_den = nullptr;
    double *d_stat = static_cast<double *>(d_stat_.ensure(sizeof(double) * 4 * D));
    const int nblk = kern::column_sums_blocks(N, D);
    double *d_part = static_cast<double *>(d_partials_f64_.ensure(sizeof(double) * nblk * 2 * D));
    if (D > 512) throw Unsupported("output_dim > 512");
    // LDS accumulators are int32 and one block adds at most `chunk_rows` rows into a cell: the power-of-two scale keeps
    // chunk_rows * max|q| < 2^31 (exactness of the wrapped int32 sums, kernels.hip k_hist_build).
    // A histogram block accumulates one chunk of one node's rows.  Chunks are sized per level so that the whole level is ONE
    // balanced round of <= 32 chunks x (feature groups) blocks (k_hist_build keeps one block per CU); `chunk_rows` is the cap
    // the fixed-point scale is derived from.  Leaf sums: int64 fixed point with n_global * max|g| * 2^lbits < 2^62.
    // It depends on the GLOBAL row count only (clamped to [4096, 65536]), so the scale -- and with it every integer sum -- is
    // the same for any sharding of the same rows.
    auto chunk_rows_of = [](long long n) { return static_cast<int>(std::min<long long>(65536, std::max<long long>(4096, 2 * ((n + 31) / 32)))); };
    int chunk_rows = chunk_rows_of(n_global);      // (row-sharded: set again once the global count is known)
    kern::StepScales *d_scales = static_cast<kern::StepScales *>(d_scales_.ensure(sizeof(kern::StepScales)));
    bool stats_fused = false;
    int32_t *d_qg = static_cast<int32_t *>(d_qg_.ensure(sizeof(int32_t) * n_el));
    // RL-sized steps on one GPU: statistics, quantisation, split candidates and class codes in ONE launch (kern::small_prep, below, where the
    // candidate buffers exist); GBRL_HIP_NO_SMALL_PREP=1 (tests / measurement): the separate launches
    const bool no_small_prep = [] { const char *e = hooks::raw(hooks::NO_SMALL_PREP); return e && e[0] == '1'; }();   /* read per call: the tests flip it */
    const bool no_small_stats = [] { const char *e = hooks::raw(hooks::NO_SMALL_STATS); return e && e[0] == '1'; }();   /* read per call: the tests flip it */
    const bool no_sort_codes = [] { const char *e = hooks::raw(hooks::SORT_NO_CODES); return e && e[0] == '1'; }();   /* read per call: the tests flip it */
    const bool uniform_gen = md.generator_type == GBRL_HIP_GEN_UNIFORM;
    const bool prep_candidate = !has_coll_ && n_global == N && F > 0 && fixed_thr_.empty() && !candidates_only_ && !no_small_prep && !no_sort_codes &&
                                !force_bisection_ && !force_sample_select_ && !force_radix_ && N <= (uniform_gen ? 8192 : kern::sort_quantiles_max_rows());
    auto run_stats = [&]() {
        // sums -> mean -> centred squares -> std, maxima, scales: all on the device (k_stats_mean / k_stats_finish); the host
        // reads the scales together with the thresholds (one synchronisation for both).  Row-sharded runs sum the column
        // sums (fp64) and take the maxima (fp32, exact) over ranks between the kernels; the arithmetic stays on the device,
        // so one GPU and N GPUs execute the same instructions on the same global sums.
        double *d_stat2 = d_stat + 2 * D;
        // one message per statistics round: the sums and every rank's maxima (gathered through the sum, kern::stats_pack)
        // (+ one word: this rank's row count in the first round -- integers below 2^53 add exactly in float64)
        const size_t smsg_words = static_cast<size_t>(D) * (world + 1) + 1;
        double *d_smsg = has_coll_ ? static_cast<double *>(d_maxbits_.ensure(sizeof(double) * smsg_words)) : nullptr;
        auto exchange_stats = [&](double *st, bool with_row_count) {
            if (!has_coll_) return;
            kern::stats_pack(st, D, world, coll_.rank, d_smsg, s, with_row_count ? static_cast<double>(N) : 0.0);
            exchange(Red::SumF64, d_smsg, smsg_words);
            kern::stats_unpack(d_smsg, D, world, st, s);
            if (with_row_count) {
                // the host needs the global count from here on (chunk length -> fixed-point scale, target ranks of the selection, leaf scale)
                double hv = 0.0;
                hip_check(hipMemcpyAsync(&hv, d_smsg + smsg_words - 1, sizeof(hv), hipMemcpyDeviceToHost, s), "D2H n");
                hip_check(hipStreamSynchronize(s), "sync");
                n_global = static_cast<long long>(hv);
                chunk_rows = chunk_rows_of(n_global);
            }
        };
        // RL-sized batch on one GPU: statistics and quantisation in ONE launch with the same reduction tree (kern::small_stats)
        if (!has_coll_ && !no_small_stats && n_global == N && kern::small_stats(dgrads, N, D, !cosine, chunk_rows, d_stat, d_meanden, d_scales, d_qg, s)) {
            stats_fused = true;
            if (!cosine) { d_mean = d_meanden; d_den = d_meanden + D; }
        } else {
        kern::column_sums(dgrads, N, D, nullptr, d_part, nblk, d_stat, s);
        exchange_stats(d_stat, true);
        if (!cosine) {
            kern::stats_mean(d_stat, n_global, D, d_meanden, s);
            kern::column_sums(dgrads, N, D, d_meanden, d_part, nblk, d_stat2, s);
            exchange_stats(d_stat2, false);
            kern::stats_finish(d_stat, d_stat2, n_global, D, chunk_rows, d_meanden, d_scales, s);
            d_mean = d_meanden;
            d_den = d_meanden + D;
        } else {
            kern::stats_finish(d_stat, nullptr, n_global, D, chunk_rows, d_meanden, d_scales, s);
        }
        }
        if (!stats_fused) kern::quantize_grads(dgrads, n_el, D, d_mean, d_den, d_scales, d_qg, s);
    };
    if (!prep_candidate) run_stats();
    phase_end("grad_stats");

    // ---- 2. split candidates ----------------------------------------------------------------------------------------
    phase_begin();
    // thresholds and scales reach the host through ONE pinned block, copied behind the binning kernel: nothing waits for them
    // until the first level's result block has arrived
    const size_t n_thr = static_cast<size_t>(F) * B;
    char *pin_ts = static_cast<char *>(pin_thr_.ensure(sizeof(float) * std::max<size_t>(1, n_thr) + sizeof(kern::StepScales) + 64));
    float *h_thr = reinterpret_cast<float *>(pin_ts);
    kern::StepScales *h_scales_pin = reinterpret_cast<kern::StepScales *>(pin_ts + ((sizeof(float) * std::max<size_t>(1, n_thr) + 63) & ~static_cast<size_t>(63)));
    float *d_thr = static_cast<float *>(d_thr_.ensure(sizeof(float) * std::max<size_t>(1, n_thr)));
    uint32_t *d_thrkeys = static_cast<uint32_t *>(d_thrkeys_.ensure(sizeof(uint32_t) * std::max<size_t>(1, n_thr)));
    uint32_t *d_kt = nullptr;
    int pass1_chunks = 0;
    if (F > 0 && !prep_candidate) {
        // order-preserving keys, feature-major: every later pass over the observations (selection, binning) streams columns
        d_kt = static_cast<uint32_t *>(d_kt_.ensure(sizeof(uint32_t) * static_cast<size_t>(N) * F));
        // quantile candidates by radix selection (the branch numeric_thresholds will take): the first digit is counted while the keys
        // pass through LDS (k_transpose_count), which saves one of the selection's passes over the keys
        const bool radix_path = fixed_thr_.empty() && md.generator_type != GBRL_HIP_GEN_UNIFORM && !force_bisection_ && !force_sample_select_ &&
                                B <= kern::radix_max_targets() && n_global < (1ll << 32);
        if (radix_path)
            pass1_chunks = kern::transpose_keys_count(dobs, N, F, d_kt, static_cast<uint32_t *>(d_radix_partial_.ensure(kern::radix_partial_bytes(F))), s);
        if (pass1_chunks == 0) kern::transpose_keys(dobs, N, F, d_kt, s);
    }
    phase_end("transpose");
    phase_begin();
    // class codes (group-major: [slot/16][row][slot%16], u16): allocated here because the RL-sized selection (one sort kernel per
    // step) writes the numeric codes itself
    const int n_slots_codes = F + Fc;
    const int n_code_groups = (n_slots_codes + kern::kCodeGroup - 1) / kern::kCodeGroup;
    const size_t code_elems = static_cast<size_t>(std::max(1, n_code_groups)) * N * kern::kCodeGroup;
    uint16_t *d_codes = static_cast<uint16_t *>(d_codes_.ensure(sizeof(uint16_t) * code_elems));
    // Only the padding slots behind the last categorical column are written by nobody (the numeric writers fill whole groups of 16, padding
    // included, the categorical ones every (row, column)): no clearing when the slots fill their groups exactly (configs[4]: 192 + 64).
    if (Fc > 0 && (F + Fc) % kern::kCodeGroup != 0) hip_check(hipMemsetAsync(d_codes, 0, sizeof(uint16_t) * code_elems, s), "memset codes");
    bool codes_from_sort = false;
    root_le_ = nullptr;
    bool prep_done = false;
    const uint16_t *d_codes_fm = nullptr;   // feature-major copy of the numeric codes (kern::small_prep writes it for kern::small_grow)
    int prep_launches = 3;   // (diagnostic, reported as a pseudo-phase at profiling level 2: 1 = the fused preparation kernel ran)
    if (prep_candidate) {
        const int64_t *d_cum = uniform_gen ? nullptr : quantile_cum_device(quantile_target_ranks(n_global, B), n_global, B);
        bool stats_done = false;
        uint16_t *d_fm = static_cast<uint16_t *>(d_codes_fm_.ensure(sizeof(uint16_t) * static_cast<size_t>(N) * F));
        prep_done = kern::small_prep(dobs, N, F, B, uniform_gen, d_cum, d_thr, d_thrkeys, d_codes, d_fm, dgrads, D, !cosine, chunk_rows, d_stat, d_meanden, d_scales, d_qg,
                                     !no_small_stats, &stats_done, s);
        if (prep_done) { codes_from_sort = true; last_quantile_fallback_ = false; prep_launches = 1; d_codes_fm = d_fm; }
        if (stats_done) stats_fused = true;
        else run_stats();
        if (!prep_done) {   // the shape did not qualify after all: the separate launches, keys first
            d_kt = static_cast<uint32_t *>(d_kt_.ensure(sizeof(uint32_t) * static_cast<size_t>(N) * F));
            kern::transpose_keys(dobs, N, F, d_kt, s);
        }
    }
    if (F > 0 && !prep_done) numeric_thresholds(dobs, N, F, B, n_global, d_kt, d_thr, d_thrkeys, pass1_chunks, d_codes, &codes_from_sort);
    phase_end("candidates");
    prof_marks_[1] = std::chrono::steady_clock::now();
    prep_launches_ = F > 0 ? prep_launches : 0;
    // numeric class codes (3. below) before the host waits for the categorical scan
    phase_begin();
    if (F > 0 && !codes_from_sort) kern::bin_cols(d_kt, N, F, d_thrkeys, B, d_codes, s);
    phase_end("binning");

    // categorical candidates (A5): distinct cells found on the device, inserted into the reference's container in the
    // reference's insertion order on the host => same candidate order (Q8).  Falls back to the host scan of every cell when
    // the batch holds more distinct categories than candidates are kept (the reference then ranks them by mean gradient norm).
    std::vector<CatCandidate> cat_cands;
    std::vector<uint16_t> h_catcodes;
    std::vector<int> cat_classes(Fc, 0);
    bool cat_codes_on_device = false;
    if (Fc > 0) {
        if (fixed_cat_valid_) {   // fit(): candidates of the whole data set (fitter.cpp:152-164); the dictionary is still on the device
            cat_cands = fixed_cat_cands_;
            cat_classes = fixed_cat_classes_;
            cat_codes_on_device = true;
        } else {
            cat_codes_on_device = (has_coll_ || !force_host_categorical_) &&
                                  device_categorical_candidates(dcells, cat_dev ? nullptr : cat, N, Fc, B, cat_cands, cat_classes);
        }
        if (!cat_codes_on_device) {
            std::vector<char> cat_host_buf;
            const char *hcat = cat;
            std::vector<float> grads_host_buf;
            const float *hgrads = grads;
            if (cat_dev) {
                cat_host_buf.resize(static_cast<size_t>(N) * Fc * kCat);
                hip_check(hipMemcpy(cat_host_buf.data(), cat, cat_host_buf.size(), hipMemcpyDeviceToHost), "D2H cat");
                hcat = cat_host_buf.data();
            }
            if (grads_dev) {
                grads_host_buf.resize(static_cast<size_t>(N) * D);
                hip_check(hipMemcpy(grads_host_buf.data(), grads, grads_host_buf.size() * 4, hipMemcpyDeviceToHost), "D2H grads");
                hgrads = grads_host_buf.data();
            }
            cat_cands.clear();
            std::fill(cat_classes.begin(), cat_classes.end(), 0);
            if (has_coll_) sharded_categorical_ranking(hcat, hgrads, N, Fc, D, B, cat_cands, h_catcodes, cat_classes);
            else categorical_candidates(hcat, hgrads, N, Fc, D, B, cat_cands, h_catcodes, cat_classes);
        }
    }

    prof_marks_[2] = std::chrono::steady_clock::now();
    if (candidates_only_) {   // fit(): only the candidates of this (whole) data set are wanted
        if (n_thr) hip_check(hipMemcpyAsync(h_thr, d_thr, sizeof(float) * n_thr, hipMemcpyDeviceToHost, s), "D2H thr");
        hip_check(hipStreamSynchronize(s), "sync");
        fixed_thr_.assign(h_thr, h_thr + n_thr);
        if (Fc > 0) {
            verify_pending_categories();
            if (cat_clash_) { cat_clash_ = false; cat_items_.clear(); cat_tab_key_.clear(); cat_tab_id_.clear(); throw HipError("two different categories of one feature share a 64-bit hash: fit() refused"); }
            if (!cat_codes_on_device)
                throw Unsupported("fit(): the data set holds more distinct categories than Fc * n_bins (mean-gradient ranking of the whole data set is not implemented)");
            fixed_cat_cands_ = cat_cands;
            fixed_cat_classes_ = cat_classes;
            fixed_cat_valid_ = true;
        }
        phases_resolve();
        return;
    }
    // ---- feature slots, candidate order, weights ---------------------------------------------------------------------
    const int n_slots = F + Fc;
    int NB = F > 0 ? B + 1 : 1;
    for (int c = 0; c < Fc; ++c) NB = std::max(NB, cat_classes[c] + 1);
    int FG = 16;
    while (FG > 1 && kern::hist_lds_bytes(NB, D, FG) > 160 * 1024 - 512) FG >>= 1;
    // more than 16 outputs: k_hist_build_wide spreads the D + 1 fields of a row over the 16 / FG parts of a 16-lane row (<= 16 each)
    while (D > 16 && FG > 4 && (16 / FG) * 16 < D + 1) FG >>= 1;
    if (kern::hist_lds_bytes(NB, D, FG) > 160 * 1024 - 512)
        throw Unsupported("(classes per feature) x (output_dim + 1) does not fit the 160 KiB LDS");
    if (static_cast<size_t>(NB + 2) * (D + 1) * 8 > 150 * 1024) throw Unsupported("score kernel LDS limit");
    const int Fp = ((n_slots + FG - 1) / FG) * FG;
    const int n_groups = Fp / FG;
    // internal candidate order = slot-grouped; cand_ref maps to the reference's candidate index (numeric f-major, then the
    // categorical candidates in the hash-map order) which decides ties (lowest reference index wins)
    std::vector<FeatureSlot> slots_local;
    std::vector<int32_t> cand_ref_local, cand_slot_local;
    std::vector<float> cand_w_local;
    std::vector<int> ref_to_internal_local;
    // numeric-only steps: these constants depend on (F, n_bins, policy, feature weights, feature mapping) only -- built once, kept in
    // Engine::step_const_ together with their uploaded copy (grow_tree)
    const bool const_cacheable = Fc == 0 && F > 0;
    // mixed steps: the numeric candidates come first in every table and are the same from step to step -- only the categorical
    // tails (this batch's candidates) are rebuilt and uploaded (51 200 numeric against ~2 000 categorical entries at configs[4])
    const bool prefix_cacheable = Fc > 0 && F > 0;
    bool reuse = false;
    if (const_cacheable || prefix_cacheable) {
        StepConstCache &cc = step_const_;
        reuse = cc.valid && cc.F == F && cc.B == B && cc.oblivious == (oblivious ? 1 : 0) && cc.fw == model.feature_weights && cc.rev == model.reverse_num;
        if (!reuse) {
            cc.valid = false;
            cc.dev_base = nullptr;
            cc.F = F; cc.B = B; cc.oblivious = oblivious ? 1 : 0;
            cc.fw = model.feature_weights;
            cc.rev = model.reverse_num;
        }
    }
    const bool shared_tables = const_cacheable || prefix_cacheable;
    std::vector<FeatureSlot> &slots = shared_tables ? step_const_.slots : slots_local;
    std::vector<int32_t> &cand_ref = shared_tables ? step_const_.cand_ref : cand_ref_local;
    std::vector<int32_t> &cand_slot = shared_tables ? step_const_.cand_slot : cand_slot_local;
    std::vector<float> &cand_w = shared_tables ? step_const_.cand_w : cand_w_local;
    std::vector<int> &ref_to_internal = shared_tables ? step_const_.ref_to_internal : ref_to_internal_local;
    int n_cand = 0;
    const int n_num_cand = F * B;
    if (reuse && const_cacheable) {
        n_cand = static_cast<int>(cand_ref.size());
    } else {
        const bool keep_prefix = reuse && prefix_cacheable;   // the numeric part of every table is in place
        slots.resize(n_slots);
        if (keep_prefix) {
            n_cand = n_num_cand;
        } else {
            for (int f = 0; f < F; ++f) { slots[f] = {0, B, n_cand, 0}; n_cand += B; }
        }
        for (int c = 0; c < Fc; ++c) { slots[F + c] = {1, cat_classes[c], n_cand, 0}; n_cand += cat_classes[c]; }
        if (keep_prefix) {
            cand_ref.resize(n_cand); cand_w.resize(n_cand); ref_to_internal.resize(n_cand); cand_slot.resize(n_cand);
        } else {
            cand_ref.assign(n_cand, 0);
            cand_w.assign(n_cand, 0.0f);
            ref_to_internal.assign(n_cand, 0);
            cand_slot.assign(n_cand, 0);
            for (int f = 0; f < F; ++f)
                for (int k = 0; k < B; ++k) {
                    const int j = slots[f].cand_base + k;
                    cand_ref[j] = f * B + k;
                    // feature weight: greedy indexes by feature_idx, oblivious by the reverse mapping (fitter.cpp:331 vs 432-434, Q6)
                    const int wi = oblivious ? model.reverse_num[f] : f;
                    cand_w[j] = (wi >= 0 && wi < md.input_dim) ? model.feature_weights[wi] : 0.0f;
                }
            for (int j = 0; j < n_num_cand; ++j) ref_to_internal[cand_ref[j]] = j;
            for (int fs = 0; fs < F; ++fs)
                for (int k = 0; k < slots[fs].n_cand; ++k) cand_slot[slots[fs].cand_base + k] = fs;
        }
        for (size_t q = 0; q < cat_cands.size(); ++q) {
            const CatCandidate &cc = cat_cands[q];
            const int j = slots[F + cc.feat].cand_base + (cc.cls - 1);
            cand_ref[j] = F * B + static_cast<int>(q);
            const int wi = oblivious ? model.reverse_cat[cc.feat] : cc.feat + F;
            cand_w[j] = (wi >= 0 && wi < md.input_dim) ? model.feature_weights[wi] : 0.0f;
        }
        for (int j = n_num_cand; j < n_cand; ++j) ref_to_internal[cand_ref[j]] = j;
        for (int fs = F; fs < n_slots; ++fs)
            for (int k = 0; k < slots[fs].n_cand; ++k) cand_slot[slots[fs].cand_base + k] = fs;
        if (shared_tables) step_const_.valid = true;
    }

    // ---- 3. class codes (group-major: [slot/16][row][slot%16], u16) ---------------------------------------------------
    if (Fc > 0) phase_begin();
    if (Fc > 0 && cat_codes_on_device) {
        if (cat_table_.valid && !fixed_cat_valid_)
            kern::cat_step_codes_table(dcells, N, Fc, F, cat_table_.keys, cat_table_.slot_q, cat_table_.cls_of_q, cat_table_.log2_cap, d_codes, s);
        else
            kern::cat_step_codes(dcells, N, Fc, F, sdict_off_, sdict_hash_, sdict_cls_, sdict_words_, d_codes, s);
    } else if (Fc > 0) {
        uint16_t *d_cc2 = static_cast<uint16_t *>(d_catcodes_.ensure(sizeof(uint16_t) * h_catcodes.size()));
        hip_check(hipMemcpyAsync(d_cc2, h_catcodes.data(), sizeof(uint16_t) * h_catcodes.size(), hipMemcpyHostToDevice, s), "H2D cat codes");
        kern::scatter_cat_codes_grouped(d_cc2, N, Fc, F, d_codes, s);
    }
    if (Fc > 0) phase_end("cat_codes");
    // thresholds and scales reach the pinned block through ONE launch (device-written host memory) instead of two copy-engine transfers:
    // kern::publish_pair at the top of the level loop (grow_tree); the one-launch growth publishes the scales itself and hands the
    // winners' thresholds over with its result blocks
    static_assert(sizeof(kern::StepScales) % 4 == 0, "copied as 32-bit words");
    void *pin_dev = nullptr;
    hip_check(hipHostGetDevicePointer(&pin_dev, pin_ts, 0), "hipHostGetDevicePointer");

    prof_marks_[3] = std::chrono::steady_clock::now();
    // ---- 4. growth (level-synchronous; Engine::grow_tree) and 5. leaf sums --------------------------------------------------
    GrowCtx gc{};
    gc.N = N; gc.F = F; gc.Fc = Fc; gc.D = D; gc.B = B; gc.MD = MD; gc.NB = NB; gc.FG = FG; gc.Fp = Fp; gc.n_groups = n_groups;
    gc.n_slots = n_slots; gc.n_cand = n_cand; gc.chunk_rows = chunk_rows; gc.n_global = n_global; gc.cosine = cosine; gc.oblivious = oblivious;
    gc.slots = &slots; gc.cand_w = &cand_w; gc.cand_ref = &cand_ref; gc.ref_to_internal = &ref_to_internal; gc.cand_slot = &cand_slot;
    gc.const_cacheable = const_cacheable; gc.cat_cands = &cat_cands;
    gc.prefix_cacheable = prefix_cacheable; gc.n_num_cand = n_num_cand; gc.cand_cap = (F + Fc) * B;
    gc.pub_thr_dev = static_cast<char *>(pin_dev); gc.pub_scales_dev = static_cast<char *>(pin_dev) + (reinterpret_cast<char *>(h_scales_pin) - pin_ts); gc.pub_thr_bytes = sizeof(float) * n_thr;
    gc.h_thr = h_thr; gc.h_scales = h_scales_pin; gc.d_thr = d_thr; gc.d_thrkeys = d_thrkeys; gc.root_le = (Fc == 0 && !has_coll_) ? root_le_ : nullptr; gc.d_kt = d_kt; gc.d_codes = d_codes; gc.d_codes_fm = d_codes_fm; gc.d_qg = d_qg; gc.dgrads = dgrads; gc.d_meanden = cosine ? nullptr : d_meanden; gc.d_scales = d_scales;
    std::vector<HNode> nodes;
    std::vector<int> frontier;
    std::vector<int64_t> acc;
    double leaf_scale = 1.0;
    grow_tree(gc, nodes, frontier, acc, leaf_scale);
    verify_pending_categories();
    if (cat_clash_) {
        // Two different cells of one feature share a 64-bit hash (2^-64 per pair of cells).  The tree just grown used the wrong
        // dictionary id for one of them, but nothing has been booked yet (append_tree below is what changes the model): forget the remembered
        // cells, switch THIS model to the host scan of every cell -- it compares bytes, so the pair cannot clash again -- and grow the tree
        // once more from the same inputs.  (Before round 6 the step threw, and every later step that held both cells threw again.)
        cat_clash_ = false;
        cat_items_.clear(); cat_tab_key_.clear(); cat_tab_id_.clear(); std::fill(cat_seen_.begin(), cat_seen_.end(), 0u);
        if (force_host_categorical_) throw HipError("two different categories of one feature share a 64-bit hash on the host scan (internal error)");
        if (has_coll_) throw HipError("two different categories of one feature share a 64-bit hash: step refused (row-sharded run: set GBRL_HIP_HOST_CATEGORICAL=1 on every rank)");
        force_host_categorical_ = true;
        ++cat_clash_redos_;
        step(obs, obs_dev, cat, cat_dev, grads, grads_dev, n, n_num, n_cat);
        return;
    }
    append_tree(model, nodes, frontier, acc, leaf_scale, cat_cands);
    (void)world;
    hip_check(hipGetLastError(), "step kernels");
    phases_resolve();
}

// ===================================================================================================== fit
float Engine::fit(const float *obs, bool obs_dev, const char *cat, bool cat_dev, const float *targets, bool targets_dev, int n, int n_num,
                  int n_cat, int iterations, bool shuffle) {
    gbrl_hip_metadata &md = model.meta;
    if (md.iteration == 0) { md.n_num_features = n_num; md.n_cat_features = n_cat; }                      // gbrl.cpp:996-999
    if (n_num != md.n_num_features || n_cat != md.n_cat_features) throw InvalidArgument("Incompatible dataset");
    if (n <= 0 || targets == nullptr) throw InvalidArgument("Cannot call fit without targets!");
    if (n_num > 0 && obs == nullptr) throw InvalidArgument("Cannot call fit without obs!");
    if (iterations < 0) throw InvalidArgument("iterations must be >= 0");
    if (n_cat > 0 && cat == nullptr) throw InvalidArgument("Cannot call fit without cat_obs!");
    if (has_coll_) throw Unsupported("fit() is not supported on a row-sharded model");
    if (md.batch_size <= 0) throw InvalidArgument("batch_size must be positive");
    ensure_device();
    hipStream_t s = stream_;
    const int F = n_num, Fc = n_cat, D = md.output_dim;
    struct Guard { Engine *e; ~Guard() { e->fixed_thr_.clear(); e->fixed_cat_valid_ = false; e->fixed_cat_cands_.clear(); e->candidates_only_ = false; } } guard{this};

    // the data set on the device, optionally in shuffled order (gbrl.cpp:1016-1024, 1039-1067; the reference seeds
    // std::mt19937 from std::random_device, i.e. the order differs from run to run there too)
    const float *dobs = obs, *dtar = targets;
    const char *dcat = cat;
    if (Fc > 0 && !cat_dev) {
        char *t = static_cast<char *>(d_fit_cells_.ensure(static_cast<size_t>(n) * Fc * kCat));
        hip_check(hipMemcpyAsync(t, cat, static_cast<size_t>(n) * Fc * kCat, hipMemcpyHostToDevice, s), "H2D cat cells");
        dcat = t;
    }
    if (F > 0 && !obs_dev) {
        float *t = static_cast<float *>(d_fit_obs_.ensure(sizeof(float) * static_cast<size_t>(n) * F));
        hip_check(hipMemcpyAsync(t, obs, sizeof(float) * static_cast<size_t>(n) * F, hipMemcpyHostToDevice, s), "H2D obs");
        dobs = t;
    }
    if (!targets_dev) {
        float *t = static_cast<float *>(d_fit_targets_.ensure(sizeof(float) * static_cast<size_t>(n) * D));
        hip_check(hipMemcpyAsync(t, targets, sizeof(float) * static_cast<size_t>(n) * D, hipMemcpyHostToDevice, s), "H2D targets");
        dtar = t;
    }
    if (shuffle) {
        std::vector<int32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0);
        std::random_device rd;
        std::mt19937 gen(rd());
        std::shuffle(perm.begin(), perm.end(), gen);
        int32_t *d_perm = static_cast<int32_t *>(d_fit_perm_.ensure(sizeof(int32_t) * n));
        hip_check(hipMemcpyAsync(d_perm, perm.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s), "H2D perm");
        float *o2 = static_cast<float *>(d_fit_obs2_.ensure(sizeof(float) * static_cast<size_t>(n) * F));
        float *t2 = static_cast<float *>(d_fit_targets2_.ensure(sizeof(float) * static_cast<size_t>(n) * D));
        kern::gather_rows(dobs, d_perm, o2, n, F, s);
        kern::gather_rows(dtar, d_perm, t2, n, D, s);
        if (Fc > 0) {   // whole 128-byte cells travel with their row (the reference's shuffled copy keeps only their first byte, Q12)
            char *c2 = static_cast<char *>(d_fit_cells2_.ensure(static_cast<size_t>(n) * Fc * kCat));
            kern::gather_rows(reinterpret_cast<const float *>(dcat), d_perm, reinterpret_cast<float *>(c2), n, Fc * (kCat / 4), s);
            dcat = c2;
        }
        hip_check(hipStreamSynchronize(s), "sync");   // perm goes out of scope
        dobs = o2;
        dtar = t2;
    }
    float *d_zero = static_cast<float *>(d_fit_zero_.ensure(sizeof(float) * D));
    hip_check(hipMemsetAsync(d_zero, 0, sizeof(float) * D, s), "memset");
    double *d_stat = static_cast<double *>(d_stat_.ensure(sizeof(double) * 4 * D));
    std::vector<double> hs(2 * D);
    auto column_stat = [&](const float *g, int rows, const float *center) {   // column sums (center null) or sums of squares
        const int nblk = kern::column_sums_blocks(rows, D);
        double *d_part = static_cast<double *>(d_partials_f64_.ensure(sizeof(double) * nblk * 2 * D));
        kern::column_sums(g, rows, D, center, d_part, nblk, d_stat, s);
        hip_check(hipMemcpyAsync(hs.data(), d_stat, sizeof(double) * 2 * D, hipMemcpyDeviceToHost, s), "D2H stat");
        hip_check(hipStreamSynchronize(s), "sync");
    };
    auto rmse = [&](const float *grads_dev, int rows) {                       // MultiRMSE, loss.cpp:42-56: sqrt(0.5 * sum g^2 / rows)
        column_stat(grads_dev, rows, d_zero);
        double tot = 0.0;
        for (int d = 0; d < D; ++d) tot += hs[d];
        return sqrtf(0.5f * static_cast<float>(tot) * (1.0f / static_cast<float>(rows)));
    };
    // bias = column means of the targets (gbrl.cpp:1075-1077)
    column_stat(dtar, n, nullptr);
    for (int d = 0; d < D; ++d) model.bias[d] = static_cast<float>(hs[d] / static_cast<double>(n));
    ++model.version;

    // split candidates from the whole data set, once (fitter.cpp:134-150)
    {
        candidates_only_ = true;
        step(dobs, true, dcat, true, dtar, true, n, F, Fc);   // returns right after the candidates; `dtar` only feeds the (unused) statistics
        candidates_only_ = false;
    }
    const int bs = md.batch_size;
    float *d_preds = static_cast<float *>(d_fit_preds_.ensure(sizeof(float) * static_cast<size_t>(std::max(n, 1)) * D));
    float *d_grads = static_cast<float *>(d_fit_grads_.ensure(sizeof(float) * static_cast<size_t>(std::min(n, bs)) * D));
    int start = 0;
    int bn = start + bs < n ? bs : n - start;                                   // fitter.cpp:120
    for (int i = 0; i < iterations; ++i) {
        const float *ob = dobs + static_cast<size_t>(start) * F;
        const float *tb = dtar + static_cast<size_t>(start) * D;
        const char *cb = Fc > 0 ? dcat + static_cast<size_t>(start) * Fc * kCat : nullptr;
        in_fit_ = true;
        try { predict(ob, true, cb, true, bn, F, Fc, 0, i, d_preds, true); } catch (...) { in_fit_ = false; throw; }   // trees [0, i) -- i == 0 means "all" (fitter.cpp:187)
        in_fit_ = false;
        kern::sub_arrays(d_preds, tb, d_grads, static_cast<size_t>(bn) * D, s);
        step(ob, true, cb, true, d_grads, true, bn, F, Fc);
        start += bn;                                                            // fitter.cpp:228-231
        if (start >= n) start = 0;
        bn = start + bs < n ? bs : n - start;
    }
    // loss on the whole data set over trees [0, iterations) (fitter.cpp:246-251)
    in_fit_ = true;   // the chain here too: the returned loss does not depend on how a stand-alone predict() would split the trees
    try { predict(dobs, true, Fc > 0 ? dcat : nullptr, true, n, F, Fc, 0, iterations, d_preds, true); } catch (...) { in_fit_ = false; throw; }
    in_fit_ = false;
    float *d_full_grads = static_cast<float *>(d_fit_grads_.ensure(sizeof(float) * static_cast<size_t>(n) * D));
    kern::sub_arrays(d_preds, dtar, d_full_grads, static_cast<size_t>(n) * D, s);
    return rmse(d_full_grads, n);
}

}  // namespace gbrl
