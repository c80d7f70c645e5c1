// engine_grow.hip -- Engine::grow_tree (see engine_step_detail.h): one tree from the prepared candidates, class codes and quantised gradients --
// RL-sized steps in one launch (k_small_grow), everything else level by level (A6-A10).  Split out of engine_step.hip in round 6; no logic changed.
#include "engine_step_detail.h"

namespace gbrl {

// ---- A6-A9, A11: level-synchronous growth of one tree from the class codes and the quantised gradients ---------------------
// Per level the host (1) uploads ONE packed descriptor block (chunk tables, slot maps, paths, partition chunks) from pinned
// memory, (2) enqueues histogram / reduce / subtract / score / argmax / resolve kernels, the read-back of ONE small result block
// (best candidate, child sizes) and -- from descriptors the device completes itself -- the partition, (3) waits for the result
// block only (an event, not the stream) and books the children while the partition runs.  Leaf sums are enqueued when a node
// becomes a leaf.  On return `nodes` is the tree, `frontier` the unsplit nodes of the last level, acc the per-node int64
// fixed-point sums of the raw gradients (| count) and leaf_scale their scale.
void Engine::grow_tree(const detail::GrowCtx &c, std::vector<detail::HNode> &nodes, std::vector<int> &frontier, std::vector<int64_t> &acc,
                       double &leaf_scale) {
    using namespace detail;
    hipStream_t s = stream_;
    const gbrl_hip_metadata &md = model.meta;
    const int N = c.N, F = c.F, D = c.D, B = c.B, MD = c.MD, NB = c.NB, FG = c.FG, Fp = c.Fp, n_groups = c.n_groups, n_slots = c.n_slots,
              n_cand = c.n_cand, chunk_rows = c.chunk_rows;
    const long long n_global = c.n_global;
    const bool cosine = c.cosine, oblivious = c.oblivious;
    const std::vector<FeatureSlot> &slots = *c.slots;
    const std::vector<float> &cand_w = *c.cand_w;
    const std::vector<int32_t> &cand_ref = *c.cand_ref;
    const std::vector<int> &ref_to_internal = *c.ref_to_internal;
    const std::vector<CatCandidate> &cat_cands = *c.cat_cands;
    const float *h_thr = c.h_thr;
    const float *d_thr = c.d_thr, *dgrads = c.dgrads;
    const uint16_t *d_codes = c.d_codes;
    const int32_t *d_qg = c.d_qg;
    kern::StepScales *d_scales = c.d_scales;
    // Level-synchronous.  Per level the host (1) uploads ONE packed descriptor block (chunk tables, slot maps, paths) from
    // pinned memory, (2) enqueues histogram / score / argmax / resolve kernels, (3) reads back ONE small result block (best
    // candidate, child sizes) -- the only synchronisation of the level -- and (4) uploads the split descriptors and enqueues
    // leaf sums and the partition without waiting for them.
    const int max_front = 1 << std::max(0, MD - 1);
    const int max_nodes = 2 * (1 << MD);
    const int max_chunks = std::max((N + 1023) / 1024, (N + kern::kPartitionRows - 1) / kern::kPartitionRows) + 2 * (1 << MD) + 2;
    const size_t n_acc = static_cast<size_t>(NB) * (D + 1) * FG;
    // RL-sized steps on one GPU grow the whole tree in ONE launch (kern::small_grow, small_grow.hip): no level buffers, no partials,
    // no row lists.  GBRL_HIP_NO_SMALL_GROW=1 (tests / measurement): the level loop below for every shape.
    const bool l2_degenerate = !c.cosine && n_global < 2;
    const bool no_small_grow = [] { const char *e = hooks::raw(hooks::NO_SMALL_GROW); return e && e[0] == '1'; }();   /* read per call: the tests flip it */
    const int small_G = (!has_coll_ && !no_small_grow && n_global == N && n_cand > 0 && !l2_degenerate && MD >= 1 && !(oblivious && device_levels_requested()) &&
                         kern::small_grow_supported(N, D, NB, MD, n_slots, n_cand)) ? kern::small_grow_blocks(n_slots) : 0;
    const bool use_small = small_G > 0 && !force_level_loop_ && !small_grow_off_;
    // per-step constants: slots, candidate weights / reference order / slot lookup
    const std::vector<int32_t> &cand_slot = *c.cand_slot;
    const size_t table_cap = c.prefix_cacheable ? static_cast<size_t>(std::max(c.cand_cap, n_cand)) : static_cast<size_t>(n_cand);
    const size_t stage_bytes = 4096 + sizeof(FeatureSlot) * n_slots + table_cap * 16 + 5 * 256 +
                               sizeof(Chunk) * (static_cast<size_t>(max_chunks) + N / 4096 + 2 * max_nodes + 64) +
                               static_cast<size_t>(max_front) * (kern::kMaxPath * 12 + 256);
    Stager stc(pin_const_, d_stage_const_, stage_bytes, s), sta(pin_a_, d_stage_a_, stage_bytes, s), stb(pin_b_, d_stage_b_, stage_bytes, s);
    FeatureSlot *d_slots;
    float *d_cand_w;
    int32_t *d_cand_ref, *d_ref_to_internal, *d_cand_slot;
    static_assert(sizeof(int) == sizeof(int32_t), "ref_to_internal is uploaded as int32");
    if (c.prefix_cacheable) {
        // fixed layout (capacity (F + Fc) * n_bins entries per table): the numeric prefixes are uploaded once per layout, every step
        // uploads the slot table and the four categorical tails
        const size_t np = static_cast<size_t>(c.n_num_cand), nt = static_cast<size_t>(n_cand) - np;
        const bool have_prefix = step_const_.dev_base == stc.device_base() && step_const_.stage_bytes == stage_bytes;
        d_slots = stc.reserve<FeatureSlot>(slots.size());
        d_cand_w = stc.reserve<float>(table_cap);
        d_cand_ref = stc.reserve<int32_t>(table_cap);
        d_ref_to_internal = stc.reserve<int32_t>(table_cap);
        d_cand_slot = stc.reserve<int32_t>(table_cap);
        char *hb = stc.host_base();
        const char *db = static_cast<const char *>(stc.device_base());
        auto mirror = [&](const void *dptr) -> char * { return hb + (static_cast<const char *>(dptr) - db); };
        void *hb_dev = nullptr;
        hip_check(hipHostGetDevicePointer(&hb_dev, hb, 0), "hipHostGetDevicePointer");
        kern::FetchSegments fs{};
        static_assert(sizeof(FeatureSlot) % 4 == 0, "fetched as 32-bit words");
        auto up = [&](void *dptr, const void *src, size_t first, size_t count, size_t elem) {   // one kernel fetches all five from the pinned mirror
            if (!count) return;
            char *hm = mirror(dptr) + first * elem;
            std::memcpy(hm, static_cast<const char *>(src) + first * elem, count * elem);
            fs.dst[fs.n] = static_cast<char *>(dptr) + first * elem;
            fs.src[fs.n] = static_cast<const char *>(hb_dev) + (hm - hb);
            fs.words[fs.n] = static_cast<uint32_t>(count * elem / 4);
            ++fs.n;
        };
        const size_t lo = have_prefix ? np : 0, cnt = have_prefix ? nt : static_cast<size_t>(n_cand);
        up(d_slots, slots.data(), 0, slots.size(), sizeof(FeatureSlot));
        up(d_cand_w, cand_w.data(), lo, cnt, 4);
        up(d_cand_ref, cand_ref.data(), lo, cnt, 4);
        up(d_ref_to_internal, ref_to_internal.data(), lo, cnt, 4);
        up(d_cand_slot, cand_slot.data(), lo, cnt, 4);
        kern::fetch_segments(fs, s);
        step_const_.dev_base = stc.device_base();
        step_const_.stage_bytes = stage_bytes;
    } else if (c.const_cacheable && step_const_.dev_base == stc.device_base() && step_const_.stage_bytes == stage_bytes) {
        d_slots = stc.reserve<FeatureSlot>(slots.size());          // uploaded by an earlier step, same layout
        d_cand_w = stc.reserve<float>(cand_w.size());
        d_cand_ref = stc.reserve<int32_t>(cand_ref.size());
        d_ref_to_internal = reinterpret_cast<int32_t *>(stc.reserve<int>(ref_to_internal.size()));
        d_cand_slot = stc.reserve<int32_t>(cand_slot.size());
    } else {
        d_slots = stc.put(slots.data(), slots.size());
        d_cand_w = stc.put(cand_w.data(), cand_w.size());
        d_cand_ref = stc.put(cand_ref.data(), cand_ref.size());
        d_ref_to_internal = reinterpret_cast<int32_t *>(stc.put(ref_to_internal.data(), ref_to_internal.size()));
        d_cand_slot = stc.put(cand_slot.data(), cand_slot.size());
        stc.flush();
        if (c.const_cacheable) { step_const_.dev_base = stc.device_base(); step_const_.stage_bytes = stage_bytes; }
    }
    int32_t *d_rows[2] = {static_cast<int32_t *>(d_rows_[0].ensure(sizeof(int32_t) * N)),
                          static_cast<int32_t *>(d_rows_[1].ensure(sizeof(int32_t) * N))};
    // A level is one balanced round of (chunks x feature groups) histogram blocks, one block per CU: 256 / n_groups chunks, at least
    // 32 (few features => more, smaller chunks; the chunk length only has an upper bound, `chunk_rows`, from the fixed-point scale).
    const int hist_chunk_budget = std::max(32, 256 / std::max(1, n_groups));
    const int hist_max_chunks = std::max(hist_chunk_budget, (N + chunk_rows - 1) / chunk_rows) + 2 * (1 << MD) + 2;
    int32_t *d_partials = static_cast<int32_t *>(d_hist_partials_.ensure(use_small ? 256 : sizeof(int32_t) * static_cast<size_t>(hist_max_chunks) * n_groups * n_acc));
    const size_t hist_node_elems = static_cast<size_t>(Fp) * NB * (D + 1);
    // two level buffers (current / previous) so that the larger child of every split can be derived as parent - sibling
    int64_t *d_hist_lvl[2] = {static_cast<int64_t *>(d_hist_.ensure(use_small ? 256 : sizeof(int64_t) * max_front * hist_node_elems)),
                              static_cast<int64_t *>(d_hist_prev_.ensure(use_small ? 256 : sizeof(int64_t) * max_front * hist_node_elems))};
    // Row-sharded runs exchange the level histograms by FEATURE: the local sums of the accumulated nodes are laid out
    // [owner rank][node][feature of the rank's slice] and reduce-scattered, so every rank receives the GLOBAL sums of its own
    // Fs = ceil(Fp / P) features only (half the bytes of an all-reduce on the xGMI ring), scores its own candidates, and the
    // level's winner is agreed with ONE small all-reduce (kern::winner_pack / winner_adopt).
    const int coll_P = has_coll_ ? std::max(1, coll_.world_size) : 1;
    const int coll_Fs = (Fp + coll_P - 1) / coll_P;                          // features per rank slice
    const int coll_lo = has_coll_ ? coll_.rank * coll_Fs : 0;                // first feature (= feature slot) of this rank
    const int own_slots = has_coll_ ? std::max(0, std::min(n_slots, coll_lo + coll_Fs) - coll_lo) : n_slots;
    const size_t feat_elems = static_cast<size_t>(NB) * (D + 1);
    int64_t *d_hist_coll = has_coll_ ? static_cast<int64_t *>(d_hist_local_.ensure(sizeof(int64_t) * max_front * static_cast<size_t>(coll_P) * coll_Fs * feat_elems)) : nullptr;
    int64_t *d_hist_recv = has_coll_ ? static_cast<int64_t *>(d_hist_recv_.ensure(sizeof(int64_t) * max_front * static_cast<size_t>(coll_Fs) * feat_elems)) : nullptr;
    int64_t *d_gather = has_coll_ ? static_cast<int64_t *>(d_gather_.ensure(sizeof(int64_t) * static_cast<size_t>(coll_P) * 3 * max_front)) : nullptr;
    // Small levels are all-reduced WHOLE instead (round 6): every rank then holds the global sums of all features, scores all candidates
    // and resolves the winner itself -- no winner exchange, no score fill, one exchange per level instead of two.  The all-reduce moves
    // twice the reduce-scatter's bytes, so it pays while that difference costs less than the winner's small all-reduce and its three
    // launches: up to ~10 MB of level payload on xGMI (8 ranks: (P-1)/P x 10 MB at ~200 GB/s bus bandwidth ~ 45 us against a ~25 us
    // all-reduce + ~20 us of launches; unmeasured beyond one GPU -- GBRL_HIP_HIST_ALLREDUCE_MAX_KB tunes it, 0 = always reduce-scatter).
    // The levels taken this way are a PREFIX of the tree: a reduce-scattered level leaves only this rank's feature slice in the level
    // buffer, and the next level's sibling subtraction reads that buffer.
    const size_t ar_max_bytes = [] { const char *e = hooks::raw(hooks::HIST_ALLREDUCE_MAX_KB); return static_cast<size_t>(e ? std::max(0L, std::atol(e)) : 10240L) * 1024; }();
    bool ar_prefix = has_coll_;      // false from the first reduce-scattered level on
    float *d_scores = static_cast<float *>(d_scores_.ensure(use_small ? 256 : sizeof(float) * static_cast<size_t>(max_front) * std::max(1, n_cand)));
    float *d_parent = static_cast<float *>(d_parent_.ensure(sizeof(float) * max_front));
    const int am_parts = kern::argmax_parts(std::max(1, n_cand));
    const size_t am_cap = static_cast<size_t>(max_front) * std::max(am_parts, std::max(1, n_slots));   // greedy: one part per feature slot
    float *d_am_v = static_cast<float *>(d_am_v_.ensure(sizeof(float) * am_cap));
    int32_t *d_am_i = static_cast<int32_t *>(d_am_i_.ensure(sizeof(int32_t) * am_cap));
    // Near-tie replay (neartie.hip; one GPU, batches of <= 65 536 rows): the selection also tracks the best DISTINCT runner-up; a node whose
    // runner-up is within `near_rel` of the winner (or whose winning gain is that close to zero) has the candidates in the window re-scored
    // in the reference's float32 sequence.  GBRL_HIP_NO_NEARTIE_REPLAY=1: the exact arg-max decides everywhere (rounds 1-4).
    const bool no_near = [] { const char *e = hooks::raw(hooks::NO_NEARTIE_REPLAY); return e && e[0] == '1'; }();   /* read per call: the tests flip it */
    const float near_rel = [] { const char *e = hooks::raw(hooks::NEARTIE_REL); return e ? static_cast<float>(std::atof(e)) : 9.5367431640625e-07f; }();   // 2^-20; measurement hook
    // Batches above 65 536 rows: at 2^20 rows x 32 768 candidates EVERY level has a runner-up inside the reference's float32 noise, and the
    // replay's chains are serial (a 2^20-row level costs 10-100 ms against a 1.85 ms step: profiles/r06_neartie_fullsize_cost.txt), so those
    // batches replay only on request -- GBRL_HIP_NEARTIE_MAX_ROWS=<n>: nodes of up to n rows (0: every node).  Unset: the exact arg-max, as in
    // rounds 1-5.  Batches of up to 65 536 rows replay every flagged node as before.
    const char *near_max_env = hooks::raw(hooks::NEARTIE_MAX_ROWS);
    const int near_max_rows = N <= kern::kNearMaxRows ? 0 : (near_max_env ? std::max(0, std::atoi(near_max_env)) : -1);   // 0: no limit, -1: no replay
    const bool near_on = !no_near && !has_coll_ && n_global == N && n_cand > 0 && kern::near_tie_supported(N, D) && near_max_rows >= 0;
    float *d_am_s = (near_on && !use_small) ? static_cast<float *>(d_am_s_.ensure(sizeof(float) * am_cap * 2)) : nullptr;
    int32_t *d_am_n = (d_am_s && N > 8192) ? reinterpret_cast<int32_t *>(d_am_s + am_cap) : nullptr;    // child sizes tell classes apart in larger batches only (score_common.h near_class)
    int32_t *d_cursors = static_cast<int32_t *>(d_cursors_.ensure(sizeof(int32_t) * max_front * 2));
    int64_t *d_leafacc = static_cast<int64_t *>(d_leafacc_.ensure(sizeof(int64_t) * max_nodes * (D + 1)));
    {   // zero unless the last tree's publication handed these words back clean
        const size_t need = sizeof(int64_t) * max_nodes * (D + 1);
        if (!use_small) {
            if (!(leafacc_clean_ptr_ == d_leafacc && need <= leafacc_clean_bytes_))
                hip_check(hipMemsetAsync(d_leafacc, 0, need, s), "memset leaf acc");
            leafacc_clean_ptr_ = nullptr;     // dirty until the end of this tree
            leafacc_clean_bytes_ = need;
        }
    }
    // result block read back once per level: [best_idx i32 x mf][best_score f32 x mf][counts i64 x 4 x mf]
    const size_t res_bytes = static_cast<size_t>(max_front) * (4 + 4 + 32) + 64;
    char *d_res = static_cast<char *>(d_results_.ensure(res_bytes));
    // The block lives in pinned host memory that the device can write: a one-block kernel publishes it (k_publish_block) and the host
    // polls a sequence word behind it -- no copy-engine launch, no event, and the partition kernel starts right behind the selection.
    char *h_res = static_cast<char *>(pin_res_.ensure(res_bytes + 64));
    volatile uint32_t *h_flag = reinterpret_cast<volatile uint32_t *>(h_res + res_bytes);
    void *h_res_dev = nullptr;
    hip_check(hipHostGetDevicePointer(&h_res_dev, h_res, 0), "hipHostGetDevicePointer");
    uint32_t *d_flag = reinterpret_cast<uint32_t *>(static_cast<char *>(h_res_dev) + res_bytes);
    *h_flag = 0;   // nothing is in flight here; a freshly allocated block must not hold a stale sequence number
    unsigned *d_pub_done = static_cast<unsigned *>(d_pub_done_.ensure(256));
    if (d_pub_done != pub_done_ptr_) {
        hip_check(hipMemsetAsync(d_pub_done, 0, 256, s), "memset publication counter");
        pub_done_ptr_ = d_pub_done;
    }
    const bool event_results = [] { const char *e = hooks::raw(hooks::EVENT_RESULTS); return e && e[0] == '1'; }();   // measurement hook
    int32_t *d_best_idx = reinterpret_cast<int32_t *>(d_res);
    float *d_best_score = reinterpret_cast<float *>(d_res + 4 * static_cast<size_t>(max_front));
    int64_t *d_counts4 = reinterpret_cast<int64_t *>(d_res + 8 * static_cast<size_t>(max_front));
    NodeSplit *d_resolved = static_cast<NodeSplit *>(d_splits_.ensure(sizeof(NodeSplit) * max_front));
    // The root's row list 0 .. N-1 is kept between steps (generated again only when N outgrows it): level 0 reads it in place of
    // d_rows[0] and, after the first partition, d_rows[0] becomes the second scratch list again.  (The device-planned loop indexes the
    // two lists by depth parity and keeps generating its own.)
    int32_t *const d_rows_b = d_rows[0];
    bool iota_root = false;
    if (!use_small) {
        // the device-planned loop partitions INTO d_rows[depth parity]: it must never be handed the cached list (same latched flag as below)
        const char *e2 = hooks::raw(hooks::NO_IOTA_CACHE);   // measurement hook
        if (!(oblivious && device_levels_requested()) && !(e2 && e2[0] == '1')) {
            int32_t *d_iota = static_cast<int32_t *>(d_rows_iota_.ensure(sizeof(int32_t) * N));
            if (d_iota != iota_ptr_ || iota_n_ < N) {
                kern::iota_rows(d_iota, N, s);
                iota_ptr_ = d_iota;
                iota_n_ = N;
            }
            d_rows[0] = d_iota;
            iota_root = true;
        } else {
            kern::iota_rows(d_rows[0], N, s);
        }
    }
    // No synchronisation here: thresholds and scales are on their way to pinned memory; the first level's event wait (or the
    // final synchronisation) covers them.  Non-finite gradients are rejected after the loop, before anything joins the model.

    nodes.clear();
    nodes.reserve(max_nodes);
    nodes.push_back(HNode{});
    nodes[0].n_local = N;
    nodes[0].n_global = n_global;
    frontier.assign(1, 0);
    int cur = 0;  // which row list is current
    std::vector<Chunk> h_chunks;
    std::vector<int32_t> h_chunk_begin;
    auto make_chunks = [&](const std::vector<int> &ids, int rows_per_chunk, bool slot_is_node_id) {
        h_chunks.clear();
        h_chunk_begin.assign(1, 0);
        for (size_t k = 0; k < ids.size(); ++k) {
            const HNode &nd = nodes[ids[k]];
            if (slot_is_node_id && nd.depth == 0) { h_chunk_begin.push_back(static_cast<int32_t>(h_chunks.size())); continue; }  // Q7
            // equal parts (no short remainder chunk): parts = ceil(n / rows_per_chunk), each ceil(n / parts) rows
            const int parts = (nd.n_local + rows_per_chunk - 1) / rows_per_chunk;
            const int each = parts ? (nd.n_local + parts - 1) / parts : 0;
            for (int off = 0; off < nd.n_local; off += each)
                h_chunks.push_back({static_cast<int32_t>(slot_is_node_id ? ids[k] : static_cast<int>(k)), nd.seg_start + off,
                                    std::min(each, nd.n_local - off), 0});
            h_chunk_begin.push_back(static_cast<int32_t>(h_chunks.size()));
        }
    };
    // smallest chunk length t (<= chunk_rows) for which the nodes `ids` need at most `budget` chunks in total
    auto balanced_chunk_rows = [&](const std::vector<int> &ids, int budget) {
        int lo = 1024, hi = chunk_rows;
        auto parts_at = [&](int t) { long long p = 0; for (int id : ids) p += (nodes[id].n_local + t - 1) / t; return p; };
        if (parts_at(hi) > budget) return hi;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (parts_at(mid) <= budget) hi = mid; else lo = mid + 1;
        }
        return hi;
    };

    // Host bookkeeping of one level from its result block [best_idx | best_score | counts]: decisions, children, paths.  Shared by the
    // level-synchronous host loop below and by the replay after a device-planned tree (one synchronisation per tree).
    struct LevelOutcome { bool stop = false; std::vector<int> splitting, new_leaves, next; };
    // (categorical feature, class) -> index into cat_cands, built at the first categorical split of the step (a linear search per
    // splitting node walked 2 000 x 136-byte records: 50 us per level at level 5 of configs[4])
    std::vector<int> cat_index, cat_index_off;
    auto cat_cand_of = [&](int feat, int cls) -> int {
        const int Fc = c.Fc;
        if (cat_index_off.empty()) {
            cat_index_off.assign(static_cast<size_t>(Fc) + 1, 0);
            for (const CatCandidate &cc : cat_cands) cat_index_off[cc.feat + 1] = std::max(cat_index_off[cc.feat + 1], cc.cls);
            for (int f = 0; f < Fc; ++f) cat_index_off[f + 1] += cat_index_off[f];
            cat_index.assign(static_cast<size_t>(cat_index_off[Fc]), -1);
            for (size_t z = 0; z < cat_cands.size(); ++z)
                if (cat_cands[z].cls >= 1) cat_index[cat_index_off[cat_cands[z].feat] + cat_cands[z].cls - 1] = static_cast<int>(z);
        }
        if (feat < 0 || feat >= Fc || cls < 1 || cls > cat_index_off[feat + 1] - cat_index_off[feat]) return -1;
        return cat_index[cat_index_off[feat] + cls - 1];
    };
    const float *win_thr = nullptr;   // small-step kernel: the winners' threshold values travel with the level's result block
    bool lazy_paths = false;          // small-step kernel: children do not copy their parent's path (in_cond[id] = the condition into node id)
    std::vector<HCond> in_cond;
    bool counts_later = false;        // small-step kernel, oblivious trees: the node sizes are derived from the leaves' row counts after the replay
    auto digest_level = [&](const std::vector<int> &active, const char *hres) -> LevelOutcome {
        LevelOutcome out;
        const int n_act = static_cast<int>(active.size());
        const int32_t *best_idx_h = reinterpret_cast<const int32_t *>(hres);
        const float *best_score_h = reinterpret_cast<const float *>(hres + 4 * static_cast<size_t>(max_front));
        const int64_t *cnt4 = reinterpret_cast<const int64_t *>(hres + 8 * static_cast<size_t>(max_front));
        const int64_t *tot_g = cnt4, *right_g = cnt4 + max_front;
        const int64_t *right_l = has_coll_ ? cnt4 + 2 * static_cast<size_t>(max_front) : right_g;
        if (oblivious && best_score_h[0] == -INFINITY) { out.stop = true; return out; }  // fitter.cpp:458
        // -- decisions (best_idx are REFERENCE candidate indices)
        std::vector<NodeSplit> sp(n_act);
        std::vector<int> &splitting = out.splitting, &new_leaves = out.new_leaves;
        for (int k = 0; k < n_act; ++k) {
            HNode &nd = nodes[active[k]];
            const int bk = oblivious ? 0 : k;
            const bool do_split = oblivious || best_score_h[bk] >= 0.0f;  // fitter.cpp:357
            NodeSplit q{};
            q.seg_start = nd.seg_start;
            if (do_split) {
                const int j = ref_to_internal[best_idx_h[bk]];
                const int fs = cand_slot[j];
                q.do_split = 1;
                q.fslot = fs;
                q.is_cat = slots[fs].is_cat;
                q.bin = slots[fs].is_cat ? (j - slots[fs].cand_base + 1) : (j - slots[fs].cand_base);
                splitting.push_back(k);
            } else {
                nd.leaf = true;
                new_leaves.push_back(active[k]);
            }
            sp[k] = q;
        }
        if (!oblivious)
            for (int id : frontier)
                if (nodes[id].n_global == 0 && !nodes[id].leaf) { nodes[id].leaf = true; new_leaves.push_back(id); }
        std::vector<int> &next = out.next;
        for (int k : splitting) {
            const int id = active[k];
            if (!counts_later && tot_g[k] != nodes[id].n_global) throw HipError("internal: histogram row count mismatch");
            const NodeSplit &q = sp[k];
            HCond c{};
            c.fslot = q.fslot;
            c.is_cat = q.is_cat != 0;
            c.bin = q.bin;
            if (c.is_cat) {
                c.feat_idx = q.fslot - F;
                c.value = INFINITY;
                c.cat_cand = cat_cand_of(c.feat_idx, q.bin);
            } else {
                c.feat_idx = q.fslot;
                c.value = win_thr ? win_thr[k] : h_thr[static_cast<size_t>(q.fslot) * B + q.bin];
                c.cat_cand = -1;
            }
            const long long npar = nodes[id].n_global, nr = right_g[k], nl = npar - nr;
            HNode l, r;
            l.depth = r.depth = nodes[id].depth + 1;
            l.parent = r.parent = id;
            HCond cl = c, cr = c;
            cl.dir = false;
            cl.edge_w = npar > 0 ? static_cast<float>(nl) / static_cast<float>(npar) : 0.0f;  // node.cpp:131
            cr.dir = true;
            cr.edge_w = npar > 0 ? static_cast<float>(nr) / static_cast<float>(npar) : 0.0f;
            if (lazy_paths) {   // (one-launch growth: only the leaves' paths are ever read -- built once, at the end, from the conditions that lead INTO the nodes)
                in_cond.resize(nodes.size() + 2);
                in_cond[nodes.size()] = cl;
                in_cond[nodes.size() + 1] = cr;
            } else {
                l.path = nodes[id].path;
                r.path = nodes[id].path;
                l.path.push_back(cl);
                r.path.push_back(cr);
            }
            const int nl_local = static_cast<int>(nodes[id].n_local - right_l[k]);
            l.seg_start = nodes[id].seg_start;
            l.n_local = nl_local;
            l.n_global = nl;
            r.seg_start = nodes[id].seg_start + nl_local;
            r.n_local = static_cast<int>(right_l[k]);
            r.n_global = nr;
            sp[k].n_left = nl_local;
            nodes[id].left = static_cast<int>(nodes.size());
            nodes.push_back(l);
            nodes[id].right = static_cast<int>(nodes.size());
            nodes.push_back(r);
            next.push_back(nodes[id].left);
            next.push_back(nodes[id].right);
        }
        return out;
    };

    // L2 with ONE row: the reference's unbiased variance is 0/0 (math_ops.cpp:461-513), every standardised gradient and every
    // split score is NaN, no comparison succeeds and the tree stays a depth-0 leaf (fitter.cpp:357, :458)
    if (!use_small) kern::publish_pair(d_thr, c.pub_thr_dev, c.pub_thr_bytes, d_scales, c.pub_scales_dev, sizeof(kern::StepScales), s);
    // ---- RL-sized steps: ONE launch grows the tree, ONE wait, then the bookkeeping is replayed from the per-level result blocks -------
    if (use_small) {
        const size_t res_stride = kern::small_grow_res_stride(MD);
        const size_t res_all = res_stride * MD;
        const size_t acc_words = static_cast<size_t>(2u << MD) * (D + 1);
        const size_t o_acc = (res_all + 255) & ~static_cast<size_t>(255), o_status = o_acc + sizeof(int64_t) * acc_words;
        char *h_blk = static_cast<char *>(pin_res_all_.ensure(o_status + 64 + 64));
        void *h_blk_dev = nullptr;
        hip_check(hipHostGetDevicePointer(&h_blk_dev, h_blk, 0), "hipHostGetDevicePointer");
        char *d_blk = static_cast<char *>(h_blk_dev);
        volatile uint32_t *h_status = reinterpret_cast<volatile uint32_t *>(h_blk + o_status);
        unsigned *d_sync = static_cast<unsigned *>(d_sg_sync_.ensure(4096));
        if (d_sync != sg_sync_ptr_) {
            hip_check(hipMemsetAsync(d_sync, 0, 4096, s), "memset barrier words");
            sg_sync_ptr_ = d_sync;
        }
        kern::SmallGrowIO io{};
        io.codes = d_codes; io.codes_fm = c.d_codes_fm; io.n_fm = c.d_codes_fm ? F : 0; io.n_thr_slots = F; io.qg = d_qg; io.grads = dgrads; io.scales = d_scales; io.slots = d_slots; io.thr = d_thr; io.cand_w = d_cand_w; io.cand_ref = d_cand_ref;
        io.N = N; io.D = D; io.B = B; io.n_slots = n_slots; io.NB = NB; io.MD = MD; io.min_data = md.min_data_in_leaf; io.cosine = cosine; io.oblivious = oblivious;
        io.G = small_G;
        io.bests = d_sg_bests_.ensure(kern::small_grow_bests_bytes(MD, small_G, oblivious));
        io.sync = d_sync;
        io.res = d_blk; io.res_dev = static_cast<char *>(d_res_all_.ensure(res_all)); io.acc = reinterpret_cast<int64_t *>(d_blk + o_acc); io.status = reinterpret_cast<uint32_t *>(d_blk + o_status);
        uint32_t seq = ++level_seq_;
        if (seq == 0) seq = ++level_seq_;
        io.seq = seq;
        io.scales_out = reinterpret_cast<kern::StepScales *>(c.pub_scales_dev);
        io.near_rel = near_on ? near_rel : 0.0f;
        if (near_on) { io.near_scratch = d_sg_near_.ensure(kern::small_grow_near_bytes(small_G, N, MD)); io.meanden = c.d_meanden; }
        const bool sg_prof = [] { const char *e = hooks::raw(hooks::SMALL_GROW_PROF); return e && e[0] == '1'; }();   // measurement hook
        if (sg_prof) io.prof = reinterpret_cast<uint32_t *>(d_blk + o_status + 64);
        h_status[0] = 0;
        // The one-launch kernel is an optimisation, never a requirement: when it cannot be launched (LDS budget, device attributes) or its
        // blocks abandon a grid barrier (they were not co-resident: another process, stream or model held CUs / LDS), nothing has been
        // booked yet -- `nodes` and `frontier` are untouched -- so the level loop grows this tree, and this engine keeps to it from now on.
        auto level_loop_instead = [&](const char *why) {
            sg_sync_ptr_ = nullptr;           // (the barrier words are in an unknown state)
            small_grow_off_ = true;
            ++small_grow_fallbacks_;
            if (md.verbose > 0) fprintf(stderr, "gbrl_hip: %s; this model grows its trees level by level from now on\n", why);
            grow_tree(c, nodes, frontier, acc, leaf_scale);
        };
        const int sg_fail = [] { const char *e = hooks::raw(hooks::TEST_SMALL_GROW_FAIL); return e ? std::atoi(e) : 0; }();   /* read per call: test hook (1: launch failure, 2: abandoned barrier) */
        phase_begin();
        if (sg_fail == 1 || !kern::small_grow(io, s)) { (void)hipGetLastError(); level_loop_instead("the one-launch growth kernel could not be launched"); return; }
        phase_end("small_grow");
        const auto t_launched = std::chrono::steady_clock::now();
        verify_pending_categories();   // (host work hidden behind the kernel)
        spin_until_published(h_status, seq, s, "small-step tree");
        const auto t_seen = std::chrono::steady_clock::now();
        hip_check(hipGetLastError(), "growth kernel");
        if (h_status[3] == 2 && !io.replay) {
            // a level of this tree has a near-tie: the kernel variant that replays a flagged node itself grows the tree once more (the
            // default variant only detects: the replay code inside it slows every step, small_grow.hip)
            io.replay = true;
            io.resume = h_status[5] == 1;     // (the default variant left its state: only the flagged level's second pass and what follows run again)
            seq = ++level_seq_;
            if (seq == 0) seq = ++level_seq_;
            io.seq = seq;
            h_status[0] = 0;
            phase_begin();
            if (!kern::small_grow(io, s)) { (void)hipGetLastError(); level_loop_instead("the one-launch growth kernel (near-tie replay variant) could not be launched"); return; }
            phase_end("small_grow");
            spin_until_published(h_status, seq, s, "small-step tree (near-tie replay)");
            hip_check(hipGetLastError(), "growth kernel");
        }
        near_in_kernel_ += h_status[4];
        if (h_status[3] == 2) {
            // a level of this tree has a near-tie: the level loop grows it, with the candidates in the window re-scored in the reference's order
            ++near_bailouts_;
            struct Reset { bool &f; ~Reset() { f = false; } } reset{force_level_loop_};
            force_level_loop_ = true;
            grow_tree(c, nodes, frontier, acc, leaf_scale);
            return;
        }
        if (h_status[3] != 0 || sg_fail == 2) {
            level_loop_instead("the one-launch growth kernel gave up at a grid barrier (its blocks were not co-resident)");
            return;
        }
        if (sg_prof) {
            static const char *names[14] = {"codes", "zero", "accumulate", "scan", "carries", "score", "select", "slot_best", "barrier", "winners", "tables", "route", "level_end", "leaves"};
            const volatile uint32_t *pw = reinterpret_cast<const volatile uint32_t *>(h_blk + o_status + 64);
            std::string line = "[small_grow block 0, us]";
            for (int i = 0; i < 14; ++i) line += std::string(" ") + names[i] + " " + std::to_string(pw[i] / 100.0).substr(0, 5);
            fprintf(stderr, "%s\n", line.c_str());
        }
        struct HostProf { bool on; std::chrono::steady_clock::time_point t0, t1, t2; ~HostProf() {
            if (!on) return;
            const auto t3 = std::chrono::steady_clock::now();
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "[small step host, us] entry->growth launched %.1f  wait %.1f  replay %.1f\n", us(t0, t1), us(t1, t2), us(t2, t3));
        } } host_prof{sg_prof, prof_step_entry_, t_launched, t_seen};
        if (sg_prof) {
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "[small step host, us] inputs+cat launch %.1f  preparation enqueued %.1f  categorical candidates (host) %.1f  tables + cat codes %.1f  grow_tree to launch %.1f\n",
                    us(prof_step_entry_, prof_marks_[0]), us(prof_marks_[0], prof_marks_[1]), us(prof_marks_[1], prof_marks_[2]), us(prof_marks_[2], prof_marks_[3]), us(prof_marks_[3], t_launched));
        }
        const int levels_written = static_cast<int>(h_status[1]);
        counts_later = oblivious;
        lazy_paths = true;
        in_cond.assign(1, HCond{});
        in_cond.reserve(static_cast<size_t>(2) << MD);
        for (int depth = 0; depth < MD; ++depth) {
            std::vector<int> active;
            for (int id : frontier)
                if (oblivious || nodes[id].n_global > 0) active.push_back(id);
            if (active.empty()) break;
            if (depth >= levels_written) throw HipError("internal: the growth kernel wrote fewer levels than the replay needs");
            const char *hres = h_blk + static_cast<size_t>(depth) * res_stride;
            win_thr = reinterpret_cast<const float *>(hres + 40 * static_cast<size_t>(max_front));
            LevelOutcome lvl = digest_level(active, hres);
            if (lvl.stop) break;
            if (lvl.splitting.empty()) { frontier.clear(); break; }
            frontier = lvl.next;
        }
        win_thr = nullptr;
        counts_later = false;
        lazy_paths = false;
        in_cond.resize(nodes.size());
        for (int id : frontier)
            if (!nodes[id].leaf) nodes[id].leaf = true;
        if (nodes.size() == 1) nodes[0].leaf = true;
        if (nodes.size() != static_cast<size_t>(h_status[2])) throw HipError("internal: the growth kernel numbered " + std::to_string(h_status[2]) + " nodes, the replay " + std::to_string(nodes.size()));
        const int64_t *h_acc = reinterpret_cast<const int64_t *>(h_blk + o_acc);
        acc.assign(nodes.size() * (D + 1), 0);
        for (size_t id = 0; id < nodes.size(); ++id)
            if (nodes[id].left < 0) std::memcpy(&acc[id * (D + 1)], h_acc + id * (D + 1), sizeof(int64_t) * (D + 1));
        if (oblivious && nodes.size() > 1) {
            // An oblivious level keeps both children of every node, so the kernel never counts them: a node's size is the sum of its leaves'
            // row counts (bottom-up: children have higher ids than their parent), and the edge weights follow (node.cpp:131).
            std::vector<long long> cnt(nodes.size(), 0);
            for (size_t id = nodes.size(); id-- > 0;)
                cnt[id] = nodes[id].left < 0 ? acc[id * (D + 1) + D] : cnt[nodes[id].left] + cnt[nodes[id].right];
            if (cnt[0] != N) throw HipError("internal: the leaves of the grown tree hold " + std::to_string(cnt[0]) + " of " + std::to_string(N) + " rows");
            for (size_t id = 0; id < nodes.size(); ++id) {
                HNode &nd = nodes[id];
                nd.n_global = cnt[id];
                nd.n_local = static_cast<int>(cnt[id]);
                if (id > 0) in_cond[id].edge_w = cnt[nd.parent] > 0 ? static_cast<float>(cnt[id]) / static_cast<float>(cnt[nd.parent]) : 0.0f;
            }
        }
        for (size_t id = 0; id < nodes.size(); ++id) {   // the leaves' paths (what append_tree writes into the model), root first
            HNode &nd = nodes[id];
            if (nd.left >= 0 || nd.depth == 0) continue;
            nd.path.resize(nd.depth);
            int at = static_cast<int>(id);
            for (int d = nd.depth - 1; d >= 0; --d) { nd.path[d] = in_cond[at]; at = nodes[at].parent; }
        }
        if (!std::isfinite(c.h_scales->hmax_build) || !std::isfinite(c.h_scales->hmax_raw)) throw InvalidArgument("non-finite gradients");
        leaf_scale = c.h_scales->leaf_scale;
        return;
    }
    // ---- oblivious trees on one GPU, opt-in (GBRL_HIP_DEVICE_LEVELS=1): the whole tree is enqueued without a host round trip per level.
    // k_plan_oblivious builds every level's descriptors on the device from the previous level's resolved splits; the consumers run
    // on worst-case grids (unused chunk entries have len 0).  The host synchronises ONCE, reads all levels' result blocks and
    // replays the bookkeeping (digest_level).  Measured (round 2, profiles/r02_device_levels.txt): the planner launch (~10 us) and the
    // empty blocks of the worst-case grids cost what the host round trip (~30 us, partly hidden behind the partition) costs --
    // 2.301 vs 2.307 ms per step at 2^20 x 128, and 0.64 vs 0.59 ms at 4096 x 128 -- so the level-synchronous host loop stays the
    // default; the test suite checks that both grow the same bytes.
    const bool host_levels = !device_levels_requested();
    const bool device_plan = oblivious && !has_coll_ && !host_levels && MD > 0 && MD <= 11 /* k_plan_oblivious: <= 1024 nodes per level */ && n_cand > 0 && !l2_degenerate;
    if (device_plan) {
        const int mf = max_front;
        const int cap_h = hist_chunk_budget + mf + 2;
        const int cap_p = (N + kern::kPartitionRows - 1) / kern::kPartitionRows + mf + 2;
        if (cap_h > hist_max_chunks) throw HipError("internal: chunk table overflow");
        // carve the plan out of one device block
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~static_cast<size_t>(255); return o; };
        const size_t o_seg = take(sizeof(int32_t) * (MD + 1) * mf), o_n = take(sizeof(int32_t) * (MD + 1) * mf);
        const size_t o_chunks = take(sizeof(Chunk) * cap_h), o_cb = take(sizeof(int32_t) * (mf + 2));
        const size_t o_sm = take(sizeof(int32_t) * mf), o_sp = take(sizeof(int32_t) * mf), o_ss = take(sizeof(int32_t) * mf);
        const size_t o_pl = take(sizeof(int32_t) * mf), o_ps = take(sizeof(int32_t) * mf * kern::kMaxPath), o_pb = take(sizeof(int32_t) * mf * kern::kMaxPath);
        const size_t o_ir = take(sizeof(int32_t) * mf), o_pv = take(sizeof(float) * mf * kern::kMaxPath);
        const size_t o_pc = take(sizeof(Chunk) * cap_p), o_st = take(sizeof(int32_t) * mf), o_state = take(sizeof(int32_t) * 4);
        const size_t o_cs = take(sizeof(int32_t) * kern::kMaxPath), o_cbin = take(sizeof(int32_t) * kern::kMaxPath), o_cv = take(sizeof(float) * kern::kMaxPath);
        char *pb_ = static_cast<char *>(d_plan_.ensure(off));
        kern::ObliviousPlan pl{};
        pl.node_seg = reinterpret_cast<int32_t *>(pb_ + o_seg); pl.node_n = reinterpret_cast<int32_t *>(pb_ + o_n); pl.mf = mf;
        pl.chunks = reinterpret_cast<Chunk *>(pb_ + o_chunks); pl.cap_h = cap_h; pl.chunk_begin = reinterpret_cast<int32_t *>(pb_ + o_cb);
        pl.slot_map = reinterpret_cast<int32_t *>(pb_ + o_sm); pl.sub_par = reinterpret_cast<int32_t *>(pb_ + o_sp); pl.sub_sib = reinterpret_cast<int32_t *>(pb_ + o_ss);
        pl.path_len = reinterpret_cast<int32_t *>(pb_ + o_pl); pl.path_slot = reinterpret_cast<int32_t *>(pb_ + o_ps); pl.path_bin = reinterpret_cast<int32_t *>(pb_ + o_pb);
        pl.is_root = reinterpret_cast<int32_t *>(pb_ + o_ir); pl.path_val = reinterpret_cast<float *>(pb_ + o_pv);
        pl.part_chunks = reinterpret_cast<Chunk *>(pb_ + o_pc); pl.cap_p = cap_p; pl.seg_starts = reinterpret_cast<int32_t *>(pb_ + o_st);
        pl.state = reinterpret_cast<int32_t *>(pb_ + o_state);
        pl.cond_slot = reinterpret_cast<int32_t *>(pb_ + o_cs); pl.cond_bin = reinterpret_cast<int32_t *>(pb_ + o_cbin); pl.cond_val = reinterpret_cast<float *>(pb_ + o_cv);
        // one result block per level
        char *d_res_all = static_cast<char *>(d_res_all_.ensure(res_bytes * MD));
        char *h_res_all = static_cast<char *>(pin_res_all_.ensure(res_bytes * MD));
        for (int depth = 0; depth < MD; ++depth) {
            const int n_act = 1 << depth, n_comp = depth == 0 ? 1 : n_act / 2;
            char *d_resL = d_res_all + static_cast<size_t>(depth) * res_bytes;
            int32_t *best_idx_L = reinterpret_cast<int32_t *>(d_resL);
            float *best_score_L = reinterpret_cast<float *>(d_resL + 4 * static_cast<size_t>(max_front));
            int64_t *counts_L = reinterpret_cast<int64_t *>(d_resL + 8 * static_cast<size_t>(max_front));
            const float *best_prev = depth ? reinterpret_cast<const float *>(d_res_all + static_cast<size_t>(depth - 1) * res_bytes + 4 * static_cast<size_t>(max_front)) : nullptr;
            int64_t *d_hist = d_hist_lvl[depth & 1];
            const int64_t *d_hist_prev = d_hist_lvl[(depth & 1) ^ 1];
            phase_begin();
            kern::plan_oblivious_level(depth, N, chunk_rows, hist_chunk_budget, depth ? d_resolved : nullptr, best_prev, d_thr, B, pl, s);
            phase_end("plan");
            {
                const auto ev = kernel_events("hist_build", /*key=*/true);
                kern::hist_build(d_codes, N, d_qg, D, d_rows[depth & 1], pl.chunks, cap_h, n_groups, FG, NB, d_partials, s, ev.first, ev.second);
            }
            phase_begin();
            kern::hist_reduce(d_partials, pl.chunk_begin, pl.slot_map, n_comp, n_groups, FG, NB, D, Fp, d_hist, s, std::max(1, hist_chunk_budget / n_comp));
            phase_end("hist_reduce");
            phase_begin();
            kern::score_candidates(d_hist, d_hist_prev, depth > 0 ? pl.sub_par : nullptr, pl.sub_sib, n_act, Fp, NB, D, d_slots, n_slots, d_thr, B, n_cand, md.min_data_in_leaf,
                                   cosine ? 1 : 0, d_scales, pl.path_len, pl.path_slot, pl.path_val, pl.path_bin, d_scores, d_parent, d_cand_w, d_cand_ref, pl.is_root,
                                   nullptr, d_am_i, s);
            kern::argmax(d_scores, n_act, n_cand, d_cand_w, d_cand_ref, d_parent, pl.is_root, true, d_am_v, d_am_i, best_idx_L, best_score_L, s);
            kern::resolve_splits(d_am_v, d_am_i, am_parts, best_idx_L, best_score_L, true, n_act, d_ref_to_internal, d_cand_slot, d_slots, d_hist, nullptr, Fp, NB, D, d_resolved,
                                 counts_L, max_front, pl.seg_starts, d_cursors, c.d_thrkeys, B, s);
            phase_end("score_select");
            phase_begin();
            kern::partition_rows(d_rows[depth & 1], d_rows[(depth & 1) ^ 1], d_codes, c.d_kt, N, pl.part_chunks, cap_p, d_resolved, d_cursors, s);
            phase_end("partition");
        }
        hip_check(hipMemcpyAsync(h_res_all, d_res_all, res_bytes * MD, hipMemcpyDeviceToHost, s), "D2H tree results");
        hip_check(hipStreamSynchronize(s), "sync tree");
        hip_check(hipGetLastError(), "growth kernels");
        // replay the bookkeeping level by level
        for (int depth = 0; depth < MD; ++depth) {
            std::vector<int> active = frontier;   // oblivious: the whole level
            if (static_cast<int>(active.size()) != (1 << depth)) throw HipError("internal: level size mismatch");
            LevelOutcome lvl = digest_level(active, h_res_all + static_cast<size_t>(depth) * res_bytes);
            if (lvl.stop || lvl.splitting.empty()) { if (!lvl.stop) frontier.clear(); break; }
            cur ^= 1;
            frontier = lvl.next;
        }
    }
    for (int depth = 0; depth < MD && n_cand > 0 && !l2_degenerate && !device_plan; ++depth) {
        // nodes that take part at this level: oblivious -> the whole level; greedy -> nodes with rows (fitter.cpp:300)
        std::vector<int> active;
        for (int id : frontier)
            if (oblivious || nodes[id].n_global > 0) active.push_back(id);
        if (active.empty()) break;
        const int n_act = static_cast<int>(active.size());
        // -- histograms.  Level 0: the root.  Deeper levels: of every sibling pair only the child with fewer rows is
        //    accumulated from the data; the other one is parent - sibling (exact integers), which halves the LDS-atomic work.
        //    (Row-sharded runs accumulate every node: the "smaller" child differs per rank.)
        //    The level buffers hold GLOBAL histograms.  Row-sharded runs pick the "smaller" child by its global row count (the
        //    same on every rank), all-reduce only those children and subtract globally.
        int64_t *d_hist = d_hist_lvl[depth & 1];
        const int64_t *d_hist_prev = d_hist_lvl[(depth & 1) ^ 1];
        std::vector<int> compute_ids;
        std::vector<int32_t> slot_map, sub_par(n_act, -1), sub_sib(n_act, -1);
        if (depth == 0) {
            compute_ids = active;
            for (int k = 0; k < n_act; ++k) slot_map.push_back(k);
        } else {
            std::vector<int> slot_of(nodes.size(), -1);
            for (int k = 0; k < n_act; ++k) slot_of[active[k]] = k;
            for (int k = 0; k < n_act; ++k) {
                const int id = active[k], par = nodes[id].parent;
                const int sib = nodes[par].left == id ? nodes[par].right : nodes[par].left;
                const bool sib_active = slot_of[sib] >= 0;
                // the child that is accumulated: fewer local rows; ties -> the left child
                const long long mine = has_coll_ ? nodes[id].n_global : nodes[id].n_local;
                const long long theirs = has_coll_ ? nodes[sib].n_global : nodes[sib].n_local;
                const bool i_am_small = sib_active && (mine < theirs || (mine == theirs && nodes[par].left == id));
                if (i_am_small) {
                    compute_ids.push_back(id);
                    slot_map.push_back(k);
                } else {
                    sub_par[k] = nodes[par].hist_slot;
                    sub_sib[k] = sib_active ? slot_of[sib] : -1;
                }
            }
        }
        for (int k = 0; k < n_act; ++k) nodes[active[k]].hist_slot = k;
        // chunk table of ALL active nodes (row-sharded runs count the local child sizes from the rows themselves)
        std::vector<Chunk> count_chunks;
        if (has_coll_) { make_chunks(active, kern::kPartitionRows, false); count_chunks = h_chunks; }
        // RL-sized levels on one GPU: every accumulated node is ONE chunk (empty nodes included) and k_hist_build stores the node's
        // int64 histogram itself -- no partials, no hist_reduce launch (kern::HistDirect).  A block then walks up to `direct_cap` rows
        // alone: the cap keeps that below ~10 us of LDS atomics ((D + 1) per row and feature).
        const bool no_direct = [] { const char *e = hooks::raw(hooks::NO_DIRECT_HIST); return e && e[0] == '1'; }();   /* read per call: the tests flip it */   // test / measurement hook
        const int direct_cap = std::min(8192, std::max(1024, 9216 / (D + 1)));
        bool hist_direct = !has_coll_ && !no_direct && kern::hist_direct_supported(FG) && !compute_ids.empty() && compute_ids.size() <= static_cast<size_t>(hist_max_chunks);
        for (int id : compute_ids) hist_direct = hist_direct && nodes[id].n_local <= direct_cap;
        if (hist_direct) {
            h_chunks.clear();
            h_chunk_begin.assign(1, 0);
            for (size_t k = 0; k < compute_ids.size(); ++k) {
                const HNode &nd = nodes[compute_ids[k]];
                h_chunks.push_back({static_cast<int32_t>(k), nd.seg_start, nd.n_local, 0});
                h_chunk_begin.push_back(static_cast<int32_t>(h_chunks.size()));
            }
        } else {
            make_chunks(compute_ids, balanced_chunk_rows(compute_ids, hist_chunk_budget), false);
        }
        if (h_chunks.size() > static_cast<size_t>(hist_max_chunks)) throw HipError("internal: chunk table overflow");
        // paths (duplicate-on-path rejection, node.cpp:154-166)
        std::vector<int32_t> pl(n_act), ps(static_cast<size_t>(n_act) * kern::kMaxPath, -1), pb(static_cast<size_t>(n_act) * kern::kMaxPath, 0), root(n_act);
        std::vector<float> pv(static_cast<size_t>(n_act) * kern::kMaxPath, 0.f);
        for (int k = 0; k < n_act; ++k) {
            const HNode &nd = nodes[active[k]];
            pl[k] = static_cast<int32_t>(nd.path.size());
            root[k] = nd.depth == 0;
            for (size_t q = 0; q < nd.path.size(); ++q) {
                ps[k * kern::kMaxPath + q] = nd.path[q].fslot;
                pv[k * kern::kMaxPath + q] = nd.path[q].value;
                pb[k * kern::kMaxPath + q] = nd.path[q].bin;
            }
        }
        sta.reset();
        Chunk *d_chunks = sta.put(h_chunks.data(), h_chunks.size());
        int32_t *d_chunk_begin = sta.put(h_chunk_begin.data(), h_chunk_begin.size());
        int32_t *d_slotmap = sta.put(slot_map.data(), slot_map.size());
        int32_t *d_sub_par = sta.put(sub_par.data(), sub_par.size());
        int32_t *d_sub_sib = sta.put(sub_sib.data(), sub_sib.size());
        int32_t *d_path_len = sta.put(pl.data(), pl.size());
        int32_t *d_path_slot = sta.put(ps.data(), ps.size());
        float *d_path_val = sta.put(pv.data(), pv.size());
        int32_t *d_path_bin = sta.put(pb.data(), pb.size());
        int32_t *d_isroot = sta.put(root.data(), root.size());
        Chunk *d_count_chunks = sta.put(count_chunks.data(), count_chunks.size());
        // One GPU: the partition of this level is enqueued right behind the selection kernels, from descriptors the device
        // completes itself (k_resolve_splits), so that it runs while the host is still waiting for / digesting the read-back.
        std::vector<Chunk> part_chunks;
        std::vector<int32_t> seg_starts(n_act), n_locals(n_act);
        {
            for (int k = 0; k < n_act; ++k) { seg_starts[k] = nodes[active[k]].seg_start; n_locals[k] = nodes[active[k]].n_local; }
            std::vector<Chunk> keep = h_chunks;
            std::vector<int32_t> keep_begin = h_chunk_begin;
            make_chunks(active, kern::kPartitionRows, false);
            part_chunks = h_chunks;
            h_chunks = keep;
            h_chunk_begin = keep_begin;
        }
        Chunk *d_part_chunks = sta.put(part_chunks.data(), part_chunks.size());
        int32_t *d_seg_starts = sta.put(seg_starts.data(), seg_starts.size());
        int32_t *d_n_locals = sta.put(n_locals.data(), n_locals.size());
        sta.flush();
        // root of a numeric-only tree on one GPU whose candidates came from the radix selection: the class counts are known from the
        // selection's ranks, so the histogram build skips the count atomic (8 instead of 9 per (row, feature) at D = 8) and hist_reduce
        // writes the counts (GBRL_HIP_ROOT_COUNTS=0: accumulate them like every other level; =2: do both and compare, the tests)
        const int root_mode = [] { const char *e = hooks::raw(hooks::ROOT_COUNTS); return e ? std::atoi(e) : 1; }();   /* read per call: the tests flip it; 2 = verify */
        const bool root_countless = depth == 0 && c.root_le != nullptr && root_mode != 0 && !hist_direct && !has_coll_ && n_global == N && NB == B + 1 &&
                                    kern::hist_countless_supported(D, FG, N);
        const bool ar_level = ar_prefix && sizeof(int64_t) * compute_ids.size() * hist_node_elems <= ar_max_bytes;
        ar_prefix = ar_level;
        const int lvl_slots = ar_level ? n_slots : own_slots, lvl_lo = (has_coll_ && !ar_level) ? coll_lo : 0;   // the feature slots this rank scores at this level
        bool hist_written = false;
        if (!h_chunks.empty()) {
            const auto ev = kernel_events("hist_build", /*key=*/true);   // the dispatch's own timestamps: no bubble in the stream
            kern::HistDirect hd;
            if (hist_direct) { hd.hist = d_hist; hd.slot_map = d_slotmap; hd.Fp = Fp; }
            hist_written = kern::hist_build(d_codes, N, d_qg, D, d_rows[cur], d_chunks, static_cast<int>(h_chunks.size()), n_groups, FG, NB, d_partials, s,
                                            ev.first, ev.second, hist_direct ? &hd : nullptr, !root_countless);
        }
        if (!hist_written) phase_begin();   // (no phase record for a level whose histograms k_hist_build stored itself)
        if (!has_coll_) {
            if (!compute_ids.empty() && !hist_written)
                kern::hist_reduce(d_partials, d_chunk_begin, d_slotmap, static_cast<int>(compute_ids.size()), n_groups, FG, NB, D, Fp, d_hist, s,
                                  static_cast<int>(h_chunks.size() / compute_ids.size()), 0, root_countless ? c.root_le : nullptr, F, B, N);
        } else if (!compute_ids.empty() && ar_level) {
            // whole-level all-reduce: plain [node][feature][class][D+1] layout (scatter with ONE owner), global sums to the level slots
            const int nc = static_cast<int>(compute_ids.size());
            bool in_place = true;      // the computed nodes fill the first level slots in order (the root; a level whose first nc nodes are the smaller children)
            for (int k = 0; k < nc; ++k) in_place = in_place && slot_map[k] == k;
            int64_t *buf = in_place ? d_hist : d_hist_coll;
            kern::hist_reduce(d_partials, d_chunk_begin, nullptr, nc, n_groups, FG, NB, D, Fp, buf, s, static_cast<int>(h_chunks.size() / nc), Fp);
            exchange(Red::SumI64, buf, static_cast<size_t>(nc) * hist_node_elems);
            if (!in_place) kern::hist_place(buf, d_hist, d_slotmap, nc, hist_node_elems, s);
        } else if (!compute_ids.empty()) {
            // local sums of the computed nodes in the feature-scattered send layout -> ONE reduce-scatter -> this rank's feature
            // slice of the global sums goes to the nodes' level slots (the other features of d_hist are never read on this rank)
            const int nc = static_cast<int>(compute_ids.size());
            if (coll_P * coll_Fs != Fp) hip_check(hipMemsetAsync(d_hist_coll, 0, sizeof(int64_t) * static_cast<size_t>(coll_P) * nc * coll_Fs * feat_elems, s), "memset");
            kern::hist_reduce(d_partials, d_chunk_begin, nullptr, nc, n_groups, FG, NB, D, Fp, d_hist_coll, s, static_cast<int>(h_chunks.size() / nc), coll_Fs);
            reduce_scatter_i64(d_hist_coll, d_hist_recv, static_cast<size_t>(nc) * coll_Fs * feat_elems);
            kern::hist_place_slice(d_hist_recv, d_hist, d_slotmap, nc, coll_Fs, coll_lo, Fp, feat_elems, s);
        }
        if (root_countless && root_mode == 2) {
            // GBRL_HIP_ROOT_COUNTS=2 (tests): the root's count fields once more by accumulation, compared entry by entry
            int64_t *d_alt = d_hist_lvl[(depth & 1) ^ 1];
            kern::hist_build(d_codes, N, d_qg, D, d_rows[cur], d_chunks, static_cast<int>(h_chunks.size()), n_groups, FG, NB, d_partials, s, nullptr, nullptr, nullptr, true);
            kern::hist_reduce(d_partials, d_chunk_begin, d_slotmap, static_cast<int>(compute_ids.size()), n_groups, FG, NB, D, Fp, d_alt, s, static_cast<int>(h_chunks.size() / compute_ids.size()));
            const size_t ne = static_cast<size_t>(Fp) * NB * (D + 1);
            std::vector<int64_t> ha(ne), hb(ne);
            hip_check(hipMemcpyAsync(ha.data(), d_hist, ne * 8, hipMemcpyDeviceToHost, s), "D2H root histogram");
            hip_check(hipMemcpyAsync(hb.data(), d_alt, ne * 8, hipMemcpyDeviceToHost, s), "D2H root histogram");
            hip_check(hipStreamSynchronize(s), "sync");
            for (int f = 0; f < F; ++f)
                for (int cl = 0; cl < NB; ++cl)
                    for (int d = 0; d <= D; ++d) {
                        const size_t i = (static_cast<size_t>(f) * NB + cl) * (D + 1) + d;
                        if (ha[i] != hb[i])
                            throw HipError("root histogram check: feature " + std::to_string(f) + " class " + std::to_string(cl) + " field " + std::to_string(d) + ": " +
                                           std::to_string(ha[i]) + " from the selection's ranks, " + std::to_string(hb[i]) + " accumulated");
                    }
        }
        if (!hist_written) phase_end("hist_reduce");
        // -- scores, selection, and the child sizes of the selected split(s): all on the device, ONE read-back
        phase_begin();
        // (row-sharded: this rank scores its own feature slots only; candidates of the other ranks stay at -inf)
        if (has_coll_ && !ar_level && oblivious) kern::fill_f32(d_scores, static_cast<size_t>(n_act) * n_cand, -INFINITY, s);
        // last level on one GPU: the derived siblings are scored but not written back (nothing subtracts from them any more)
        const bool skip_hook = [] { const char *e = hooks::raw(hooks::KEEP_LAST_DERIVED); return e && e[0] == '1'; }();   // measurement hook
        const bool drop_derived = !has_coll_ && !skip_hook && depth > 0 && depth == MD - 1;
        if (lvl_slots > 0)
            kern::score_candidates(d_hist, d_hist_prev, depth > 0 ? d_sub_par : nullptr, d_sub_sib, n_act, Fp, NB, D, d_slots, lvl_slots, d_thr, B, n_cand, md.min_data_in_leaf, cosine ? 1 : 0,
                                   d_scales, d_path_len, d_path_slot, d_path_val, d_path_bin, d_scores, d_parent, d_cand_w, d_cand_ref, d_isroot,
                                   oblivious ? nullptr : d_am_v, d_am_i, s, lvl_lo, !drop_derived, oblivious ? nullptr : d_am_s, oblivious ? nullptr : d_am_n);
        // oblivious: the scores are summed over the level's nodes first (stage 1 below); greedy: k_score has already reduced every
        // feature of every node to its best gain, so only the final reduction inside k_resolve_splits is left
        if (oblivious)
            kern::argmax(d_scores, n_act, n_cand, d_cand_w, d_cand_ref, d_parent, d_isroot, oblivious, d_am_v, d_am_i, d_best_idx, d_best_score, s, d_am_s);
        // counts4 = [total | right] from the (global) histogram; sharded runs add [right_local] counted from the local rows
        // (one GPU: the kernel itself mirrors the result block into the pinned host copy and its last block publishes the sequence word)
        uint32_t seq = 0;
        const bool publish_in_resolve = !has_coll_ && !event_results;
        if (!event_results) {
            seq = ++level_seq_;
            if (seq == 0) seq = ++level_seq_;
        }
        const bool near_level = d_am_s != nullptr && publish_in_resolve;
        const kern::NearDetect near_detect{d_am_s, oblivious ? nullptr : d_am_n, near_rel, d_parent, d_isroot, cosine ? 1 : 0, N};
        kern::resolve_splits(d_am_v, d_am_i, oblivious ? am_parts : lvl_slots, d_best_idx, d_best_score, oblivious, n_act, d_ref_to_internal, d_cand_slot, d_slots, d_hist, nullptr, Fp, NB, D, d_resolved,
                             d_counts4, max_front, d_seg_starts, d_cursors, c.d_thrkeys, B, s, publish_in_resolve ? h_res_dev : nullptr, d_flag, seq, d_pub_done,
                             drop_derived ? d_hist_prev : nullptr, drop_derived ? d_sub_par : nullptr, drop_derived ? d_sub_sib : nullptr, near_level ? &near_detect : nullptr);
        if (has_coll_) {
            if (!ar_level) {
                // the level's winner over all ranks: every rank holds the best of ITS features and the child sizes it induces
                const int n_win = oblivious ? 1 : n_act;
                const size_t gwords = static_cast<size_t>(coll_P) * (n_win + 2 * n_act);
                kern::winner_pack(d_best_idx, d_best_score, d_counts4, max_front, n_win, n_act, coll_.rank, d_gather, s, coll_P);
                exchange(Red::SumI64, d_gather, gwords);
                kern::winner_adopt(d_gather, coll_P, n_win, n_act, oblivious, d_ref_to_internal, d_cand_slot, d_slots, d_seg_starts, c.d_thrkeys, B, d_best_idx, d_best_score,
                                   d_counts4, max_front, d_resolved, d_cursors, s);
            }
            // (whole-level all-reduce: k_resolve_splits has resolved the global winner on every rank and cleared the third counts array)
            int64_t *d_right_local = d_counts4 + 2 * static_cast<size_t>(max_front);   // (cleared by winner_adopt)
            if (!count_chunks.empty())
                kern::count_right(d_rows[cur], d_codes, c.d_kt, N, d_count_chunks, static_cast<int>(count_chunks.size()), d_resolved, d_right_local, s);
            // global left sizes -> this rank's, and the completed result block to the host: one launch.  (Round 6: the counting kernel's last
            // block doing this instead cost 8 us per level MORE -- its 256 blocks queue on one completion counter, ~30 ns per returning atomic.)
            if (event_results) kern::localize_splits(d_resolved, d_n_locals, d_right_local, n_act, s);
            else kern::localize_publish(d_resolved, d_n_locals, d_right_local, n_act, d_res, h_res_dev, res_bytes, d_flag, seq, s);
        }
        if (event_results) hip_check(hipMemcpyAsync(h_res, d_res, res_bytes, hipMemcpyDeviceToHost, s), "D2H level results");
        phase_end("score_select");
        {
            if (event_results) hip_check(hipEventRecord(ev_level_, s), "hipEventRecord");
            phase_begin();
            if (!part_chunks.empty())
                kern::partition_rows(d_rows[cur], d_rows[cur ^ 1], d_codes, c.d_kt, N, d_part_chunks, static_cast<int>(part_chunks.size()), d_resolved,
                                     d_cursors, s);
            phase_end("partition");
            // spin on the event (a blocking wait costs a thread wake-up of ~10-20 us per level; the wait itself is a few tens of us)
            if (event_results) {
                for (;;) {
                    const hipError_t q = hipEventQuery(ev_level_);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) hip_check(q, "hipEventQuery(level results)");
                }
            } else {
                // poll the sequence word; now and then ask the stream for errors (a faulted kernel would never publish)
                verify_pending_categories();   // (first level only does work: hidden behind the level's kernels)
                spin_until_published(h_flag, seq, s, "level results");
            }
        }
        hip_check(hipGetLastError(), "growth kernels");
        if (near_level) {
            // flags of the level (k_resolve_splits): any -> the candidates in the window are scored once more, the reference's way, the final
            // arg-max stage runs on their outcome and the partition -- already enqueued from the exact decision, its input list is intact -- runs again
            const int64_t *near_h = reinterpret_cast<const int64_t *>(h_res + 8 * static_cast<size_t>(max_front)) + 2 * static_cast<size_t>(max_front);
            // nodes above the requested size limit keep the exact arg-max (GBRL_HIP_NEARTIE_MAX_ROWS, batches above 65 536 rows only; 0 = no limit;
            // an oblivious level is replayed only when every one of its nodes is within the limit)
            bool any = false;
            if (oblivious) {
                any = near_h[0] != 0;
                if (any && near_max_rows > 0) for (int k = 0; k < n_act; ++k) any = any && nodes[active[k]].n_local <= near_max_rows;
            } else {
                for (int k = 0; k < n_act; ++k) any = any || (near_h[k] != 0 && (near_max_rows == 0 || nodes[active[k]].n_local <= near_max_rows));
            }
            const bool near_debug = [] { const char *e = hooks::raw(hooks::NEARTIE_DEBUG); return e && e[0] == '1'; }();   // measurement hook
            if (any && near_debug) {
                const float *bs = reinterpret_cast<const float *>(h_res + 4 * static_cast<size_t>(max_front));
                for (int k = 0; k < (oblivious ? 1 : n_act); ++k)
                    if (near_h[k]) {
                        const int32_t sb = static_cast<int32_t>(near_h[max_front + k]);
                        float sec; std::memcpy(&sec, &sb, 4);
                        fprintf(stderr, "[near-tie] depth %d node %d of %d (%d rows): best gain %.9g (candidate %d), runner-up %.9g, difference %.3g\n", depth, k, n_act, nodes[active[k]].n_local,
                                bs[k], reinterpret_cast<const int32_t *>(h_res)[k], sec, bs[k] - sec);
                    }
            }
            if (any) {
                ++near_replays_;
                phase_begin();
                int32_t *d_cand_nr = (oblivious || N <= 8192) ? nullptr : static_cast<int32_t *>(d_near_nr_.ensure(sizeof(int32_t) * static_cast<size_t>(max_front) * std::max(1, n_cand)));
                if (!oblivious)   // every candidate's exact score and child sizes (the greedy selection kept the per-slot bests only)
                    kern::score_candidates(d_hist, d_hist_prev, depth > 0 ? d_sub_par : nullptr, d_sub_sib, n_act, Fp, NB, D, d_slots, own_slots, d_thr, B, n_cand, md.min_data_in_leaf, cosine ? 1 : 0,
                                           d_scales, d_path_len, d_path_slot, d_path_val, d_path_bin, d_scores, d_parent, d_cand_w, d_cand_ref, d_isroot, nullptr, d_am_i, s, 0, !drop_derived, nullptr, nullptr, d_cand_nr);
                kern::NearTieIO io{};
                io.rows = d_rows[cur]; io.seg_start = d_seg_starts; io.n_rows = d_n_locals; io.codes = d_codes; io.N = N; io.D = D; io.grads = dgrads; io.meanden = c.d_meanden;
                io.cosine = cosine ? 1 : 0; io.oblivious = oblivious ? 1 : 0; io.min_data = md.min_data_in_leaf; io.slots = d_slots; io.cand_slot = d_cand_slot; io.cand_w = d_cand_w; io.cand_ref = d_cand_ref;
                io.n_cand = n_cand; io.scores = d_scores; io.cand_nr = d_cand_nr; io.parent = d_parent; io.is_root = d_isroot; io.best_score = d_best_score; io.near = d_counts4 + 2 * static_cast<size_t>(max_front);
                io.rel = near_rel; io.n_act = n_act;
                int32_t *lists = static_cast<int32_t *>(d_near_list_.ensure(sizeof(int32_t) * static_cast<size_t>(max_front) * (kern::kNearCands + 1)));
                io.list = lists; io.list_n = lists + static_cast<size_t>(max_front) * kern::kNearCands;
                io.ent = static_cast<int32_t *>(d_near_ent_.ensure(sizeof(int32_t) * std::max(static_cast<size_t>(kern::kNearCands + 1) * N, static_cast<size_t>(n_cand))));
                io.rep = static_cast<float *>(d_near_rep_.ensure(sizeof(float) * static_cast<size_t>(max_front) * (kern::kNearCands + 1)));
                io.part_v = d_am_v; io.part_i = d_am_i; io.n_parts = oblivious ? am_parts : own_slots;
                io.max_node_rows = near_max_rows;
                if (const size_t mw = kern::near_tie_map_words(N, n_act)) io.maps = static_cast<uint32_t *>(d_near_maps_.ensure(sizeof(uint32_t) * mw));
                int near_largest = 0;      // the largest node this replay will walk
                for (int k = 0; k < n_act; ++k)
                    if ((oblivious || near_h[k] != 0) && (near_max_rows == 0 || nodes[active[k]].n_local <= near_max_rows)) near_largest = std::max(near_largest, nodes[active[k]].n_local);
                // (below ~10^5 rows per node the one-lane-per-chain core is the faster one: the parallel evaluation summarises 17 N D elements
                // per pass whatever the nodes' sizes -- profiles/r06_neartie_fullsize_cost.txt)
                if (kern::near_tie_fast_supported(N, D) && (reinterpret_cast<uintptr_t>(dgrads) & 15) == 0 /* float4 pieces of the gradient rows */ && !hooks::on(hooks::NEARTIE_SERIAL) && near_largest > (cosine ? 32768 : 98304)) {   // (the dot chains of Cosine are D times longer: the parallel evaluation pays off earlier)
                    // big batch, D a multiple of 4: the float32 chains are evaluated by seqsum.hip on the whole GPU (GBRL_HIP_NEARTIE_SERIAL=1: the
                    // one-lane-per-chain core of neartie_core.h, same bits -- the tests compare the two)
                    const size_t rows17 = static_cast<size_t>(kern::kNearCands + 1) * N, blocks17 = static_cast<size_t>(n_act) * (kern::kNearCands + 1);
                    io.fast = 1;
                    io.pos = static_cast<int32_t *>(d_near_pos_.ensure(sizeof(int32_t) * rows17));
                    io.nr = static_cast<int32_t *>(d_near_nrb_.ensure(sizeof(int32_t) * blocks17));
                    io.vals = static_cast<float *>(d_near_vals_.ensure(sizeof(float) * rows17 * D));
                    io.means = static_cast<float *>(d_near_means_.ensure(sizeof(float) * blocks17 * 2 * D));
                    io.sums = static_cast<float *>(d_near_sums_.ensure(sizeof(float) * blocks17 * 2 * D));
                    io.rowsort = static_cast<int32_t *>(d_near_rowsort_.ensure(sizeof(int32_t) * static_cast<size_t>(N)));
                    io.tiles = static_cast<int32_t *>(d_near_tiles_.ensure(sizeof(int32_t) * kern::near_tie_fast_tiles(N, n_act)));
                    io.seq_blocks = kern::near_tie_fast_blocks(N, D, n_act);
                    io.chains_bytes = kern::near_tie_fast_chain_bytes(N, D, n_act);
                    io.chains = d_near_chains_.ensure(io.chains_bytes);
                }
                kern::near_tie_replay(io, s);
                seq = ++level_seq_;
                if (seq == 0) seq = ++level_seq_;
                kern::resolve_splits(d_am_v, d_am_i, oblivious ? am_parts : own_slots, d_best_idx, d_best_score, oblivious, n_act, d_ref_to_internal, d_cand_slot, d_slots, d_hist, nullptr, Fp, NB, D, d_resolved,
                                     d_counts4, max_front, d_seg_starts, d_cursors, c.d_thrkeys, B, s, h_res_dev, d_flag, seq, d_pub_done,
                                     drop_derived ? d_hist_prev : nullptr, drop_derived ? d_sub_par : nullptr, drop_derived ? d_sub_sib : nullptr, nullptr);
                if (!part_chunks.empty())
                    kern::partition_rows(d_rows[cur], d_rows[cur ^ 1], d_codes, c.d_kt, N, d_part_chunks, static_cast<int>(part_chunks.size()), d_resolved, d_cursors, s);
                phase_end("near_tie_replay");
                spin_until_published(h_flag, seq, s, "level results after the near-tie replay");
                hip_check(hipGetLastError(), "near-tie replay kernels");
            }
        }
        LevelOutcome lvl = digest_level(active, h_res);
        if (lvl.stop) break;
        std::vector<int> &splitting = lvl.splitting, &new_leaves = lvl.new_leaves, &next = lvl.next;
        // -- leaves finalised at this level (their segment is intact in the current list) and the partition: enqueued, not awaited
        stb.reset();
        if (!new_leaves.empty()) {
            make_chunks(new_leaves, 1024, true);
            if (!h_chunks.empty()) {
                Chunk *d_lc = stb.put(h_chunks.data(), h_chunks.size());
                stb.flush();
                phase_begin();
                kern::leaf_sums(dgrads, D, d_rows[cur], d_lc, static_cast<int>(h_chunks.size()), d_scales, d_leafacc, s);
                phase_end("leaves");
            }
        }
        if (splitting.empty()) { frontier.clear(); break; }
        cur ^= 1;   // the partition was enqueued from the device-side descriptors (same decisions: best_score rule, n_left)
        if (iota_root) { d_rows[0] = d_rows_b; iota_root = false; }   // the root list is read-only: the next partition writes the scratch list
        frontier = next;
    }

    // ---- 5. leaves ---------------------------------------------------------------------------------------------------
    {
        std::vector<int> last;
        for (int id : frontier)
            if (!nodes[id].leaf) { nodes[id].leaf = true; last.push_back(id); }
        if (nodes.size() == 1) nodes[0].leaf = true;
        make_chunks(last, 1024, true);
        if (!h_chunks.empty()) {
            // stage B may still be in flight for the partition of the last level: stage A is free (its level is complete)
            sta.reset();
            Chunk *d_lc = sta.put(h_chunks.data(), h_chunks.size());
            sta.flush();
            phase_begin();
            kern::leaf_sums(dgrads, D, d_rows[cur], d_lc, static_cast<int>(h_chunks.size()), d_scales, d_leafacc, s);
            phase_end("leaves");
        }
    }
    if (has_coll_) {
        exchange(Red::SumI64, d_leafacc, static_cast<size_t>(nodes.size()) * (D + 1));
    }
    const size_t n_acc_words = nodes.size() * (D + 1);
    // The leaf sums reach the host the way the level results do: a one-block kernel stores them into pinned, device-mapped memory and
    // then a sequence word; the host polls it instead of a copy-engine transfer + hipStreamSynchronize (a blocking wait costs a thread
    // wake-up).  Seeing the word means every earlier operation of the stream -- all kernels that read the caller's inputs, the
    // copies of thresholds and scales -- has completed.
    const size_t acc_bytes = sizeof(int64_t) * std::max<size_t>(1, n_acc_words);
    char *h_acc_raw = static_cast<char *>(pin_acc_.ensure(acc_bytes + 64));
    int64_t *h_acc = reinterpret_cast<int64_t *>(h_acc_raw);
    if (event_results) {
        hip_check(hipMemcpyAsync(h_acc, d_leafacc, sizeof(int64_t) * n_acc_words, hipMemcpyDeviceToHost, s), "D2H leaf acc");
        hip_check(hipStreamSynchronize(s), "sync");
    } else {
        void *h_acc_dev = nullptr;
        hip_check(hipHostGetDevicePointer(&h_acc_dev, h_acc_raw, 0), "hipHostGetDevicePointer");
        volatile uint32_t *h_aflag = reinterpret_cast<volatile uint32_t *>(h_acc_raw + acc_bytes);
        uint32_t seq = ++level_seq_;
        if (seq == 0) seq = ++level_seq_;
        *h_aflag = 0;
        kern::publish_block(d_leafacc, h_acc_dev, sizeof(int64_t) * n_acc_words, reinterpret_cast<uint32_t *>(static_cast<char *>(h_acc_dev) + acc_bytes), seq, s,
                            /*zero_src=*/true);
        spin_until_published(h_aflag, seq, s, "leaf sums");
        leafacc_clean_ptr_ = d_leafacc;   // only the copied words were ever written, and the kernel cleared them
    }
    acc.assign(h_acc, h_acc + n_acc_words);
    // everything enqueued for this tree has completed: scales are in pinned memory
    if (!std::isfinite(c.h_scales->hmax_build) || !std::isfinite(c.h_scales->hmax_raw)) throw InvalidArgument("non-finite gradients");
    leaf_scale = c.h_scales->leaf_scale;

}

}  // namespace gbrl
