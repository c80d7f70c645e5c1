// engine_candidates.hip -- the split candidates of a step (see engine_step_detail.h): distinct categorical cells on the device and their
// host replay (A5), the row-sharded mean-gradient ranking, exact quantile / uniform thresholds (A3, A4), and the host side of the
// copy-free hand-overs.  Split out of engine_step.hip in round 6; no logic changed.
#include "engine_step_detail.h"

namespace gbrl {

namespace detail {

// Host side of the copy-free hand-overs: poll a sequence word in coherent pinned memory; every 16384 polls ask the stream for errors
// (a faulted kernel never publishes) and give up after GBRL_HIP_SPIN_SECONDS (default 120) of wall clock -- a hung kernel must not
// spin a core forever, and a slow but healthy run (counter profiling, several ranks sharing one device) must not be declared dead.
// Before giving up the stream is synchronised: kernels still in flight would otherwise keep storing into the pinned result blocks and
// pools that the next call reuses; a stream that does drain turns the timeout into an ordinary completion.
void spin_until_published(volatile uint32_t *flag, uint32_t seq, hipStream_t s, const char *what) {
    const double kSpinSeconds = [] { const char *e = hooks::raw(hooks::SPIN_SECONDS); const double v = e ? std::atof(e) : 0.0; return v > 0.0 ? v : 120.0; }();
    int idle = 0;
    std::chrono::steady_clock::time_point t0;
    bool timed = false;
    for (unsigned spins = 1; *flag != seq; ++spins) {
        if ((spins & 0x3fff) == 0) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) { if (++idle > 64) throw HipError(std::string("internal: ") + what + " were not published"); }
            else if (q != hipErrorNotReady) hip_check(q, what);
            if (!timed) { t0 = std::chrono::steady_clock::now(); timed = true; }
            else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kSpinSeconds) {
                hip_check(hipStreamSynchronize(s), what);   // nothing may still be writing when the caller unwinds
                if (*flag == seq) break;
                throw HipError(std::string("timeout: ") + what + " did not arrive (stream drained, nothing published)");
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
}

// split_candidate_generator.cpp:216-249: n_bins+1 equal-count buckets, threshold i = value at rank cum_i - 1.  With fewer rows than buckets
// the remainder loop still gives the first n_samples buckets one row each, so cum_i = min(i + 1, n_samples) >= 1: the ranks repeat at the
// column maximum (the reference grows valid trees there).
std::vector<int64_t> quantile_target_ranks(long long n_global, int B) {
    std::vector<int64_t> cum(B);
    const long long per = n_global / (B + 1), rem = n_global % (B + 1);
    long long run = 0;
    for (int i = 0; i < B; ++i) { run += per + (i < rem ? 1 : 0); cum[i] = run; }
    return cum;
}

}  // namespace detail


void Engine::sharded_categorical_ranking(const char *hcat, const float *hgrads, int N, int Fc, int D, int B,
                                         std::vector<detail::CatCandidate> &cat_cands, std::vector<uint16_t> &h_catcodes, std::vector<int> &cat_classes) {
    hipStream_t s = stream_;
    const int world = coll_.world_size, rank = coll_.rank;
    std::vector<float> norms(N, 0.0f);
    for (int i = 0; i < N; ++i) {   // calculate_squared_norm (math_ops.cpp:726-749), contracted like the reference build
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) { const float g = hgrads[static_cast<size_t>(i) * D + d]; acc = fmaf(g, g, acc); }
        norms[i] = acc;
    }
    // (1) local scan: local id of every cell, distinct pairs in local first-occurrence order (feature-major)
    std::unordered_map<std::string, int> local_id;
    std::vector<int> l_feat;
    std::vector<std::string> l_name;
    std::vector<int32_t> cell_lid(static_cast<size_t>(N) * Fc);
    for (int f = 0; f < Fc; ++f)
        for (int i = 0; i < N; ++i) {
            std::string name(hcat + (static_cast<size_t>(i) * Fc + f) * kCat, kCat);
            auto it = local_id.emplace(name + "_" + std::to_string(f), static_cast<int>(l_feat.size()));
            if (it.second) { l_feat.push_back(f); l_name.push_back(std::move(name)); }
            cell_lid[static_cast<size_t>(i) * Fc + f] = it.first->second;
        }
    // all-gather helper through the sum exchange: every rank writes its block into a zeroed buffer
    auto all_gather_i64 = [&](const std::vector<int64_t> &mine, std::vector<long long> &counts) -> std::vector<int64_t> {
        int64_t *d_cnt = static_cast<int64_t *>(d_cat_xchg_.ensure(sizeof(int64_t) * (world + 1)));
        std::vector<int64_t> cnt(world + 1, 0);
        cnt[rank] = static_cast<int64_t>(mine.size());
        hip_check(hipMemcpyAsync(d_cnt, cnt.data(), sizeof(int64_t) * (world + 1), hipMemcpyHostToDevice, s), "H2D");
        hip_check(hipStreamSynchronize(s), "sync");
        exchange(Red::SumI64, d_cnt, world + 1);
        hip_check(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int64_t) * (world + 1), hipMemcpyDeviceToHost, s), "D2H");
        hip_check(hipStreamSynchronize(s), "sync");
        counts.assign(cnt.begin(), cnt.begin() + world);
        size_t total = 0, off = 0;
        for (int r = 0; r < world; ++r) { if (r < rank) off += static_cast<size_t>(cnt[r]); total += static_cast<size_t>(cnt[r]); }
        if (total > (size_t(1) << 24)) throw Unsupported("too many distinct categories for a row-sharded step");
        std::vector<int64_t> all(std::max<size_t>(total, 1), 0);
        std::copy(mine.begin(), mine.end(), all.begin() + static_cast<long>(off));
        int64_t *d_all = static_cast<int64_t *>(d_cat_xchg_.ensure(sizeof(int64_t) * all.size()));
        hip_check(hipMemcpyAsync(d_all, all.data(), sizeof(int64_t) * all.size(), hipMemcpyHostToDevice, s), "H2D");
        hip_check(hipStreamSynchronize(s), "sync");
        exchange(Red::SumI64, d_all, all.size());
        hip_check(hipMemcpyAsync(all.data(), d_all, sizeof(int64_t) * all.size(), hipMemcpyDeviceToHost, s), "D2H");
        hip_check(hipStreamSynchronize(s), "sync");
        all.resize(total);
        return all;
    };
    // (2) global list of distinct pairs: 17-word records (feature, the 128 bytes) in rank order
    std::vector<int64_t> mine(l_feat.size() * 17, 0);
    for (size_t q = 0; q < l_feat.size(); ++q) {
        mine[q * 17] = l_feat[q];
        std::memcpy(&mine[q * 17 + 1], l_name[q].data(), kCat);
    }
    std::vector<long long> rec_counts;
    const std::vector<int64_t> all = all_gather_i64(mine, rec_counts);
    const size_t n_rec = all.size() / 17;
    // the reference inserts feature-major, then in row order: stable sort of the rank-major list by feature
    std::vector<int> order(n_rec);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return all[static_cast<size_t>(a) * 17] < all[static_cast<size_t>(b) * 17]; });
    struct Info { float total = 0.f; long long count = 0; int feat = 0; std::string name; int gid = -1; };
    std::unordered_map<std::string, Info> uniq;   // same container, same insertion sequence as the reference => same iteration order (Q8)
    std::vector<std::string> gkey;                // global id -> key
    for (int q : order) {
        const int f = static_cast<int>(all[static_cast<size_t>(q) * 17]);
        std::string name(reinterpret_cast<const char *>(&all[static_cast<size_t>(q) * 17 + 1]), kCat);
        std::string key = name + "_" + std::to_string(f);
        auto it = uniq.find(key);
        if (it == uniq.end()) {
            Info ci;
            ci.feat = f; ci.name = std::move(name); ci.gid = static_cast<int>(gkey.size());
            gkey.push_back(key);
            uniq.emplace(std::move(key), std::move(ci));
        }
    }
    const size_t G = gkey.size();
    std::vector<int> lid_to_gid(l_feat.size());
    for (size_t q = 0; q < l_feat.size(); ++q) lid_to_gid[q] = uniq[l_name[q] + "_" + std::to_string(l_feat[q])].gid;
    // counts: exact integer all-reduce
    std::vector<int64_t> cnts(std::max<size_t>(G, 1), 0);
    for (size_t c = 0; c < cell_lid.size(); ++c) ++cnts[lid_to_gid[cell_lid[c]]];
    {
        int64_t *d = static_cast<int64_t *>(d_cat_xchg_.ensure(sizeof(int64_t) * cnts.size()));
        hip_check(hipMemcpyAsync(d, cnts.data(), sizeof(int64_t) * cnts.size(), hipMemcpyHostToDevice, s), "H2D");
        hip_check(hipStreamSynchronize(s), "sync");
        exchange(Red::SumI64, d, cnts.size());
        hip_check(hipMemcpyAsync(cnts.data(), d, sizeof(int64_t) * cnts.size(), hipMemcpyDeviceToHost, s), "D2H");
        hip_check(hipStreamSynchronize(s), "sync");
    }
    // (3) totals: float32 sums in GLOBAL row order, one rank at a time.  The reference's loop is feature-major over ALL rows, but a key
    // belongs to one feature, so per key the order of its additions is simply the global row order.
    std::vector<double> tot(std::max<size_t>(G, 1), 0.0);   // transported as doubles (exact for float32 values), summed with zeros
    for (int r = 0; r < world; ++r) {
        std::vector<double> send(tot.size(), 0.0);
        if (r == rank) {
            std::vector<float> t32(tot.size());
            for (size_t k = 0; k < tot.size(); ++k) t32[k] = static_cast<float>(tot[k]);
            for (int f = 0; f < Fc; ++f)
                for (int i = 0; i < N; ++i) {
                    float &t = t32[lid_to_gid[cell_lid[static_cast<size_t>(i) * Fc + f]]];
                    t += norms[i];
                }
            for (size_t k = 0; k < tot.size(); ++k) send[k] = static_cast<double>(t32[k]);
        }
        double *d = static_cast<double *>(d_cat_xchg_.ensure(sizeof(double) * send.size()));
        hip_check(hipMemcpyAsync(d, send.data(), sizeof(double) * send.size(), hipMemcpyHostToDevice, s), "H2D");
        hip_check(hipStreamSynchronize(s), "sync");
        exchange(Red::SumF64, d, send.size());
        hip_check(hipMemcpyAsync(tot.data(), d, sizeof(double) * send.size(), hipMemcpyDeviceToHost, s), "D2H");
        hip_check(hipStreamSynchronize(s), "sync");
    }
    for (auto &kv : uniq) { kv.second.total = static_cast<float>(tot[kv.second.gid]); kv.second.count = cnts[kv.second.gid]; }
    // (4) the reference's ranking (split_candidate_generator.cpp:131-161)
    std::vector<std::pair<std::string, float>> vec;
    for (const auto &kv : uniq) vec.emplace_back(kv.first, kv.second.total / static_cast<float>(static_cast<int>(kv.second.count)));
    int n_unique = static_cast<int>(vec.size());
    if (n_unique > Fc * B) {
        std::sort(vec.begin(), vec.end(), [](const std::pair<std::string, float> &a, const std::pair<std::string, float> &b) { return a.second > b.second; });
        n_unique = Fc * B;
    }
    std::vector<int> cls_of_gid(std::max<size_t>(G, 1), 0);
    for (int i = 0; i < n_unique; ++i) {
        const Info &ci = uniq[vec[i].first];
        const int cls = ++cat_classes[ci.feat];
        if (cls > 65534) throw Unsupported("more than 65534 candidate categories in one feature");
        cat_cands.emplace_back(ci.feat, ci.name.data(), cls);
        cls_of_gid[ci.gid] = cls;
    }
    h_catcodes.assign(static_cast<size_t>(N) * Fc, 0);
    for (size_t c = 0; c < cell_lid.size(); ++c) h_catcodes[c] = static_cast<uint16_t>(cls_of_gid[lid_to_gid[cell_lid[c]]]);
}

// Two stages (round 4): `launch_only` enqueues the scan (tables, insert, verify, publish) -- step() calls it BEFORE the gradient
// statistics, the numeric candidates and the numeric binning, none of which depend on it -- and the second call polls the publish
// kernel's own completion word, so the host's replay of the reference's container (~0.1 ms at configs[4]) runs while the device works through
// the numeric preparation instead of in front of an idle device (0.17 ms per 4096-row step, profiles/r04_cfg5_timeline_*.txt).
bool Engine::device_categorical_candidates(const char *dcells, const char *hcells, int N, int Fc, int B,
                                           std::vector<detail::CatCandidate> &cat_cands, std::vector<int> &cat_classes, bool launch_only) {
    (void)hcells;   // the distinct cells are gathered from the device copy either way
    hipStream_t s = stream_;
    const long long keep = static_cast<long long>(Fc) * B;
    if (keep > (1 << 20)) return false;
    int full_log2 = 8;
    while ((1ll << full_log2) < 4 * std::min<long long>(N, keep + 1) && full_log2 < 20) ++full_log2;
    if ((static_cast<size_t>(Fc) << full_log2) >= (1ull << 31)) return false;   // list records are 32-bit table slots
    const bool resume = cat_launched_;   // the first round of the loop below is already on the stream
    cat_launched_ = false;
    // The per-feature tables are sized for the worst case (every row a new category: 4 N slots); real columns hold a few dozen
    // categories, so the step starts with four times the largest distinct count the previous step saw and repeats with the full size
    // only if a table overflowed (12 MB of memsets and atomics on a 12 MB table -> 0.2 MB at configs[4]).
    int log2_cap = std::min(full_log2, std::max(8, cat_log2_hint_));
    const int list_cap = static_cast<int>(keep) + 1;
    int32_t *d_meta = static_cast<int32_t *>(d_cat_meta_.ensure(sizeof(int32_t) * 4));               // flags[2], counter
    int32_t *d_lslot = static_cast<int32_t *>(d_cat_lslot_.ensure(sizeof(int32_t) * list_cap));
    uint64_t *d_keys = nullptr;
    int32_t *d_first = nullptr, *d_slotq = nullptr;
    // ONE launch writes header + records + the distinct cells themselves into mapped pinned memory, ONE synchronisation reads them
    // (round 2: three copies of lists sized by a count that needed its own round trip, then a gather + a fourth copy: four
    // synchronisations per step).  The record count is guessed from the last step; a larger batch of distinct cells is published
    // again with the exact count.
    const int32_t *h_hdr = nullptr, *lfeat = nullptr, *lfirst = nullptr;
    const uint64_t *lhash = nullptr;
    const char *names = nullptr;
    bool names_in_pinned = false;
    auto publish = [&](int cap, bool launch, bool collect) {
        const size_t bytes = 64 + static_cast<size_t>(cap) * (8 + 4 + 4 + kCat);
        char *h = static_cast<char *>(pin_cat_.ensure(bytes));
        void *dv = nullptr;
        hip_check(hipHostGetDevicePointer(&dv, h, 0), "hipHostGetDevicePointer");
        char *d = static_cast<char *>(dv);
        const size_t o_hash = 64, o_feat = o_hash + 8 * static_cast<size_t>(cap), o_first = o_feat + 4 * static_cast<size_t>(cap),
                     o_names = o_first + 4 * static_cast<size_t>(cap);   // 64 + 16 cap: 16-byte aligned
        volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(h) + 4;   // header word 4: written last, by the last block
        if (launch) {
            *flag = 0;
            d_slotq = static_cast<int32_t *>(d_cat_slotq_.ensure(sizeof(int32_t) * (static_cast<size_t>(Fc) << log2_cap)));
            kern::cat_publish(d_meta, d_lslot, d_keys, d_first, log2_cap, dcells, Fc, cap, reinterpret_cast<int32_t *>(d), reinterpret_cast<int32_t *>(d + o_feat),
                              reinterpret_cast<int32_t *>(d + o_first), reinterpret_cast<uint64_t *>(d + o_hash), d + o_names, d_slotq, ++cat_pub_seq_, s);
        }
        if (!collect) return;
        spin_until_published(flag, cat_pub_seq_, s, "the batch's distinct categorical cells");   // the publish only: kernels enqueued behind it keep running
        // the device wrote these lines over PCIe, so every first touch by the host misses its caches: ONE sequential pass (prefetcher
        // friendly) into ordinary memory, sized by the published count, instead of the replay's scattered reads (3x slower measured)
        // (Round 5: the 128-byte cells -- 260 KiB of the 290 at configs[4] -- are NOT copied on one GPU: a cell the engine has met before is
        // recognised by its 64-bit hash and feature, and its bytes are compared with the remembered ones later, while the device grows the
        // tree (verify_pending_categories); only new cells are read here.)
        const bool copy_names = has_coll_;
        const int n_pub = std::max(0, std::min(reinterpret_cast<const int32_t *>(h)[3], cap));
        cat_host_.resize(64 + static_cast<size_t>(n_pub) * (8 + 4 + 4 + (copy_names ? kCat : 0)));
        char *c = cat_host_.data();
        std::memcpy(c, h, 64);
        const size_t c_hash = 64, c_feat = c_hash + 8 * static_cast<size_t>(n_pub), c_first = c_feat + 4 * static_cast<size_t>(n_pub),
                     c_names = c_first + 4 * static_cast<size_t>(n_pub);
        std::memcpy(c + c_hash, h + o_hash, 8 * static_cast<size_t>(n_pub));
        std::memcpy(c + c_feat, h + o_feat, 4 * static_cast<size_t>(n_pub));
        std::memcpy(c + c_first, h + o_first, 4 * static_cast<size_t>(n_pub));
        if (copy_names) std::memcpy(c + c_names, h + o_names, static_cast<size_t>(kCat) * n_pub);
        h_hdr = reinterpret_cast<const int32_t *>(c);
        lhash = reinterpret_cast<const uint64_t *>(c + c_hash);
        lfeat = reinterpret_cast<const int32_t *>(c + c_feat);
        lfirst = reinterpret_cast<const int32_t *>(c + c_first);
        names = copy_names ? c + c_names : h + o_names;
        names_in_pinned = !copy_names;
    };
    for (bool first = true;; first = false) {
        const size_t slots = static_cast<size_t>(Fc) << log2_cap;
        d_keys = static_cast<uint64_t *>(d_cat_keys_.ensure(sizeof(uint64_t) * slots));
        d_first = static_cast<int32_t *>(d_cat_first_.ensure(sizeof(int32_t) * slots));
        const bool enqueued = first && resume;   // (the hints that size this round are only updated by the collecting call)
        if (enqueued) d_slotq = static_cast<int32_t *>(d_cat_slotq_.ensure(sizeof(int32_t) * slots));
        if (!enqueued) {
            if (slots <= (size_t(1) << 22)) {   // the usual few-KiB tables: one launch clears all three
                kern::FillSegments fz{};
                fz.n = 3;
                fz.dst[0] = d_keys; fz.words[0] = static_cast<uint32_t>(2 * slots); fz.value[0] = 0u;
                fz.dst[1] = d_first; fz.words[1] = static_cast<uint32_t>(slots); fz.value[1] = 0x7f7f7f7fu;
                fz.dst[2] = d_meta; fz.words[2] = 4; fz.value[2] = 0u;
                kern::fill_segments(fz, s);
            } else {
                hip_check(hipMemsetAsync(d_keys, 0, sizeof(uint64_t) * slots, s), "memset");
                hip_check(hipMemsetAsync(d_first, 0x7f, sizeof(int32_t) * slots, s), "memset");
                hip_check(hipMemsetAsync(d_meta, 0, sizeof(int32_t) * 4, s), "memset");
            }
            kern::cat_distinct_insert(dcells, N, Fc, d_keys, d_first, log2_cap, d_meta, d_lslot, d_meta + 2, list_cap, s);
            kern::cat_distinct_verify(dcells, N, Fc, d_keys, d_first, log2_cap, d_meta, s);
        }
        publish(std::min(list_cap, std::max(256, cat_publish_guess_)), !enqueued, !launch_only);
        if (launch_only) { cat_launched_ = true; return true; }
        if (h_hdr[0] != 0 && log2_cap < full_log2) { log2_cap = full_log2; continue; }   // a table (or the list) overflowed: once more at full size
        break;
    }
    const bool cat_prof = [] { const char *e = hooks::raw(hooks::CAT_PROF); return e && e[0] == '1'; }();   // measurement hook
    std::chrono::steady_clock::time_point cp[6];
    if (cat_prof) cp[0] = std::chrono::steady_clock::now();
    int n_distinct = h_hdr[2];
    bool declined = h_hdr[0] != 0 || h_hdr[1] != 0 || n_distinct > keep;
    if (has_coll_) {   // every rank must take the same path
        int64_t *d_flag = static_cast<int64_t *>(d_cat_xchg_.ensure(sizeof(int64_t)));
        int64_t hv = declined ? 1 : 0;
        hip_check(hipMemcpyAsync(d_flag, &hv, sizeof(hv), hipMemcpyHostToDevice, s), "H2D");
        exchange(Red::SumI64, d_flag, 1);
        hip_check(hipMemcpyAsync(&hv, d_flag, sizeof(hv), hipMemcpyDeviceToHost, s), "D2H");
        hip_check(hipStreamSynchronize(s), "sync");
        declined = hv != 0;
    }
    if (declined) return false;
    if (n_distinct > h_hdr[3]) publish(n_distinct, true, true);
    cat_publish_guess_ = n_distinct + n_distinct / 4 + 64;
    // the reference's insertion order: feature-major, then row of first occurrence -- one LSD radix sort (11-bit digits) of
    // feature * N + first row with the list index in the low 21 bits (std::sort of the per-feature buckets: 30 us at configs[4])
    std::vector<int> order(n_distinct);
    {
        std::vector<int> per_feat(Fc, 0);
        for (int q = 0; q < n_distinct; ++q) ++per_feat[lfeat[q]];
        int mx = 1;
        for (int f = 0; f < Fc; ++f) mx = std::max(mx, per_feat[f]);
        int l2 = 8;
        while ((1 << l2) < 4 * mx && l2 < 20) ++l2;
        cat_log2_hint_ = l2;                       // table size the next step starts with
        if (n_distinct > (1 << 21)) throw Unsupported("more than 2^21 distinct categorical cells in one step");   // (the list index rides in the key's low 21 bits; Fc * n_bins <= 2^20 above)
        std::vector<uint64_t> ka(n_distinct), kb(n_distinct);
        for (int q = 0; q < n_distinct; ++q)
            ka[q] = ((static_cast<uint64_t>(lfeat[q]) * static_cast<uint64_t>(N) + static_cast<uint64_t>(lfirst[q])) << 21) | static_cast<uint64_t>(q);
        int key_bits = 1;
        while (key_bits < 43 && (static_cast<uint64_t>(Fc) * static_cast<uint64_t>(N)) >> key_bits) ++key_bits;
        for (int sh = 21; sh < 21 + key_bits; sh += 11) {
            uint32_t cnt[2049] = {0};
            for (int q = 0; q < n_distinct; ++q) ++cnt[((ka[q] >> sh) & 2047u) + 1];
            for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
            for (int q = 0; q < n_distinct; ++q) kb[cnt[(ka[q] >> sh) & 2047u]++] = ka[q];
            ka.swap(kb);
        }
        for (int q = 0; q < n_distinct; ++q) order[q] = static_cast<int>(ka[q] & ((1u << 21) - 1));
    }
    std::vector<int32_t> g_feat;     // row-sharded: the global lists replace the local views
    std::vector<uint64_t> g_hash;
    std::vector<char> g_names;
    if (has_coll_) {
        // Row-sharded: every rank needs the distinct cells of ALL ranks, in the order a single process would meet them (rank
        // after rank = global row order).  All-gather through the sum exchange: counts first, then 18-word records
        // (feature, first row, the 128 bytes) written into rank-indexed slots of a zeroed buffer.
        const int world = coll_.world_size, rank = coll_.rank;
        int64_t *d_cnt = static_cast<int64_t *>(d_cat_xchg_.ensure(sizeof(int64_t) * (world + 1)));
        std::vector<int64_t> cnt(world + 1, 0);
        cnt[rank] = n_distinct;
        hip_check(hipMemcpyAsync(d_cnt, cnt.data(), sizeof(int64_t) * (world + 1), hipMemcpyHostToDevice, s), "H2D");
        exchange(Red::SumI64, d_cnt, world + 1);
        hip_check(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int64_t) * (world + 1), hipMemcpyDeviceToHost, s), "D2H");
        hip_check(hipStreamSynchronize(s), "sync");
        long long total = 0, my_off = 0;
        for (int r = 0; r < world; ++r) { if (r < rank) my_off += cnt[r]; total += cnt[r]; }
        if (total > (1ll << 20)) throw Unsupported("too many distinct categories for a row-sharded step");
        std::vector<int64_t> rec(static_cast<size_t>(total) * 18, 0);
        for (int q = 0; q < n_distinct; ++q) {
            int64_t *r18 = &rec[(static_cast<size_t>(my_off) + q) * 18];
            r18[0] = lfeat[order[q]];
            r18[1] = lfirst[order[q]];
            std::memcpy(r18 + 2, names + static_cast<size_t>(order[q]) * kCat, kCat);
        }
        if (total > 0) {
            int64_t *d_rec = static_cast<int64_t *>(d_cat_xchg_.ensure(sizeof(int64_t) * rec.size()));
            hip_check(hipMemcpyAsync(d_rec, rec.data(), sizeof(int64_t) * rec.size(), hipMemcpyHostToDevice, s), "H2D");
            exchange(Red::SumI64, d_rec, rec.size());
            hip_check(hipMemcpyAsync(rec.data(), d_rec, sizeof(int64_t) * rec.size(), hipMemcpyDeviceToHost, s), "D2H");
            hip_check(hipStreamSynchronize(s), "sync");
        }
        // global list, already rank-major and (feature, first row)-sorted inside a rank: stable sort by feature keeps that order
        n_distinct = static_cast<int>(total);
        g_feat.resize(n_distinct); g_hash.resize(n_distinct);
        g_names.resize(static_cast<size_t>(n_distinct) * kCat);
        for (int q = 0; q < n_distinct; ++q) {
            const int64_t *r18 = &rec[static_cast<size_t>(q) * 18];
            g_feat[q] = static_cast<int32_t>(r18[0]);
            std::memcpy(&g_names[static_cast<size_t>(q) * kCat], r18 + 2, kCat);
            uint64_t w[16];
            std::memcpy(w, r18 + 2, kCat);
            g_hash[q] = cat_cell_hash_raw(w);
        }
        lfeat = g_feat.data(); lhash = g_hash.data(); names = g_names.data(); lfirst = nullptr;
        order.resize(n_distinct);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return lfeat[a] < lfeat[b2]; });
    }
    if (cat_prof) cp[1] = std::chrono::steady_clock::now();
    // Replay of the reference's candidate container (std::unordered_map<std::string, ...> keyed by cell + "_" + feature, filled in
    // the order above, split_candidate_generator.cpp:117-130): its ITERATION order is the candidate order (Q8).  The order of a
    // libstdc++ hash table is a function of the keys' hash values and of the insertion sequence only, so the replay inserts small
    // the keys' std::hash values -- computed once per distinct (feature, cell) the engine has ever met and kept in cat_items_ --
    // instead of building and hashing 130-byte strings every step.
    // GBRL_HIP_CAT_CHECK=1 (tests) replays the string-keyed container beside it and compares the two orders.
    if (cat_items_.size() > (1u << 18)) { cat_items_.clear(); cat_tab_key_.clear(); cat_tab_id_.clear(); std::fill(cat_seen_.begin(), cat_seen_.end(), 0u); }
    // (raw hash, feature) -> head of the chain through CatItem::next: open addressing, linear probing, at most half full
    // (round 4: std::unordered_map cost 2 000 node lookups = 40 us per 4096-row step of configs[4])
    auto tab_slot = [&](uint64_t key) -> size_t {
        const size_t mask = cat_tab_key_.size() - 1;
        size_t i = static_cast<size_t>(key ^ (key >> 29)) & mask;
        while (cat_tab_id_[i] >= 0 && cat_tab_key_[i] != key) i = (i + 1) & mask;
        return i;
    };
    auto tab_reserve = [&](size_t n_items) {
        if (!cat_tab_key_.empty() && 2 * n_items <= cat_tab_key_.size()) return;
        size_t cap = 4096;
        while (cap < 4 * n_items) cap <<= 1;
        std::vector<uint64_t> ok; std::vector<int32_t> oi;
        ok.swap(cat_tab_key_); oi.swap(cat_tab_id_);
        cat_tab_key_.assign(cap, 0); cat_tab_id_.assign(cap, -1);
        for (size_t i = 0; i < ok.size(); ++i)
            if (oi[i] >= 0) { const size_t j = tab_slot(ok[i]); cat_tab_key_[j] = ok[i]; cat_tab_id_[j] = oi[i]; }
    };
    tab_reserve(cat_items_.size() + static_cast<size_t>(n_distinct));
    const bool defer_compare = names_in_pinned;
    cat_pending_.clear();
    std::vector<int> item_of_q(static_cast<size_t>(std::max(1, n_distinct)), -1);
    auto item_of = [&](int feat, uint64_t h, const char *cell) -> int {
        const uint64_t key = h * 0x9E3779B97F4A7C15ull + static_cast<uint64_t>(feat);
        const size_t slot = tab_slot(key);
        const int head = cat_tab_id_[slot];
        if (defer_compare) {
            // exactly one remembered cell with this (feature, hash): take it and compare the bytes later (verify_pending_categories);
            // several (two different cells that share a 64-bit hash have been met): compare now
            int hit = -1, hits = 0;
            for (int id = head; id >= 0; id = cat_items_[id].next)
                if (cat_items_[id].feat == feat && cat_items_[id].lhash == h) { hit = id; ++hits; }
            if (hits == 1) { cat_pending_.emplace_back(hit, cell); return hit; }
        }
        for (int id = head; id >= 0; id = cat_items_[id].next) {
            const detail::CatItem &ci = cat_items_[id];
            if (ci.feat == feat && std::memcmp(ci.name, cell, kCat) == 0) return id;
        }
        detail::CatItem ci;
        ci.feat = feat;
        ci.lhash = h;
        std::memcpy(ci.name, cell, kCat);
        std::string ks(cell, kCat);
        ks += "_" + std::to_string(feat);
        ci.std_hash = std::hash<std::string>{}(ks);
        ci.next = head;
        const int id = static_cast<int>(cat_items_.size());
        cat_items_.push_back(ci);
        cat_tab_key_[slot] = key;
        cat_tab_id_[slot] = id;
        return id;
    };
    // The replay itself: hash_order_replay.h (libstdc++'s unique-key insertion restated on index arrays).
    std::vector<int> cand_item;   // distinct-list index of every candidate, in candidate order
    cand_item.reserve(n_distinct);
    {
        std::vector<size_t> hcode;
        std::vector<int> node_q;
        hcode.reserve(n_distinct); node_q.reserve(n_distinct);
        const uint32_t tag = ++cat_seen_tag_;
        if (tag == 0) { std::fill(cat_seen_.begin(), cat_seen_.end(), 0u); cat_seen_tag_ = 1; }
        for (int q : order) {
            const int id = item_of(lfeat[q], lhash[q], names + static_cast<size_t>(q) * kCat);
            item_of_q[q] = id;
            if (static_cast<size_t>(id) >= cat_seen_.size()) cat_seen_.resize(std::max<size_t>(2 * cat_seen_.size(), static_cast<size_t>(id) + 1), 0);
            if (cat_seen_[id] == cat_seen_tag_) continue;      // key already in the container (row-sharded lists): emplace() finds it, inserts nothing
            cat_seen_[id] = cat_seen_tag_;
            hcode.push_back(cat_items_[id].std_hash);
            node_q.push_back(q);
        }
        if (cat_prof) cp[2] = std::chrono::steady_clock::now();
        for (int k : libstdcxx_unique_insert_order(hcode)) cand_item.push_back(node_q[k]);   // the container's iteration order (Q8)
        if (cat_prof) cp[3] = std::chrono::steady_clock::now();
    }
    // The replay leans on libstdc++ internals.  Production processes check it against the real container on their FIRST categorical
    // steps (eight of them: the early ones have the fewest rehashes) and then trust it; GBRL_HIP_CAT_CHECK=1 (the test suite) checks
    // every step, =0 never.  A disagreement is an error, not a silent reordering of the candidates.
    const int check_mode = [] { const char *e = hooks::raw(hooks::CAT_CHECK); return e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2)) : 2; }();
    static std::atomic<int> checks_left{8};
    const bool check_replay = check_mode == 1 || (check_mode == 2 && !order.empty() && checks_left.load(std::memory_order_relaxed) > 0 &&
                                                   checks_left.fetch_sub(1, std::memory_order_relaxed) > 0);
    if (check_replay) {
        std::unordered_map<std::string, int> ref_map;
        for (int q : order) {
            std::string key(names + static_cast<size_t>(q) * kCat, kCat);
            key += "_" + std::to_string(lfeat[q]);
            ref_map.emplace(std::move(key), q);
        }
        size_t k = 0;
        bool same = ref_map.size() == cand_item.size();
        for (const auto &kv : ref_map) { if (!same) break; same = cand_item[k++] == kv.second; }
        if (!same) throw HipError("categorical candidates: the hash-replay order differs from the string-keyed container's");
    }
    if (static_cast<long long>(cand_item.size()) > keep)
        throw Unsupported("more distinct categories than Fc * n_bins in a row-sharded step (the reference's mean-gradient ranking is not available sharded)");
    // candidates + the step's dictionary (per feature: entries sorted by raw hash, then class), packed into ONE pinned block and
    // uploaded with one copy that nothing waits for: the next write of the block happens behind the next step's synchronisation
    const int n_ent = static_cast<int>(cand_item.size());
    cat_table_.valid = false;
    if (!has_coll_ && !candidates_only_) {
        // ordinary step on one GPU: the scan's tables ARE the dictionary; the host hands back the class of every list record only
        int32_t *hc = static_cast<int32_t *>(pin_cat_cls_.ensure(sizeof(int32_t) * static_cast<size_t>(std::max(1, n_distinct))));
        cat_cands.reserve(cat_cands.size() + n_ent);
        for (int q : cand_item) {
            const int f = lfeat[q];
            const int cls = ++cat_classes[f];
            if (cls > 65534) throw Unsupported("more than 65534 candidate categories in one feature");
            cat_cands.emplace_back(f, cat_items_[item_of_q[q]].name, cls);   // (the remembered bytes: equal to the published cell's, verified below / later)
            hc[q] = cls;
        }
        int32_t *dc = static_cast<int32_t *>(d_cat_clsq_.ensure(sizeof(int32_t) * static_cast<size_t>(std::max(1, n_distinct))));
        void *hc_dev = nullptr;
        hip_check(hipHostGetDevicePointer(&hc_dev, hc, 0), "hipHostGetDevicePointer");
        kern::FetchSegments fs{};
        fs.n = 1; fs.dst[0] = dc; fs.src[0] = hc_dev; fs.words[0] = static_cast<uint32_t>(std::max(1, n_distinct));
        kern::fetch_segments(fs, s);
        cat_table_.valid = true; cat_table_.keys = d_keys; cat_table_.slot_q = d_slotq; cat_table_.cls_of_q = dc; cat_table_.log2_cap = log2_cap;
        if (cat_prof) {
            cp[4] = std::chrono::steady_clock::now();
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "[cat host, us] %d distinct: insertion order %.1f  items %.1f  container order %.1f  candidates + classes %.1f\n", n_distinct, us(cp[0], cp[1]), us(cp[1], cp[2]), us(cp[2], cp[3]), us(cp[3], cp[4]));
        }
        return true;
    }
    struct DictE { uint64_t h; int cls; int item; };
    std::vector<DictE> ent(n_ent);
    std::vector<int32_t> off(Fc + 1, 0);
    for (int q : cand_item) ++off[lfeat[q] + 1];
    for (int f = 0; f < Fc; ++f) off[f + 1] += off[f];
    {
        std::vector<int32_t> cur(off.begin(), off.end() - 1);
        cat_cands.reserve(cat_cands.size() + n_ent);
        for (int q : cand_item) {
            const int f = lfeat[q];
            const int cls = ++cat_classes[f];
            if (cls > 65534) throw Unsupported("more than 65534 candidate categories in one feature");
            cat_cands.emplace_back(f, cat_items_[item_of_q[q]].name, cls);
            ent[cur[f]++] = {lhash[q], cls, q};
        }
    }
    for (int f = 0; f < Fc; ++f)
        std::sort(ent.begin() + off[f], ent.begin() + off[f + 1], [](const DictE &a, const DictE &b2) { return a.h < b2.h || (a.h == b2.h && a.cls < b2.cls); });
    const size_t n1 = static_cast<size_t>(n_ent) + 1;   // one zero entry behind the last: the arrays are never empty
    const size_t o_words = 0, o_hash = o_words + n1 * kCat, o_off = o_hash + n1 * 8, o_cls = o_off + (static_cast<size_t>(Fc) + 1) * 4,
                 dict_bytes = o_cls + n1 * 4;
    char *hd = static_cast<char *>(pin_cat_dict_.ensure((dict_bytes + 3) & ~static_cast<size_t>(3)));
    for (int e = 0; e < n_ent; ++e) {
        std::memcpy(hd + o_words + static_cast<size_t>(e) * kCat, names + static_cast<size_t>(ent[e].item) * kCat, kCat);
        reinterpret_cast<uint64_t *>(hd + o_hash)[e] = ent[e].h;
        reinterpret_cast<int32_t *>(hd + o_cls)[e] = ent[e].cls;
    }
    std::memset(hd + o_words + static_cast<size_t>(n_ent) * kCat, 0, kCat);
    reinterpret_cast<uint64_t *>(hd + o_hash)[n_ent] = 0;
    reinterpret_cast<int32_t *>(hd + o_cls)[n_ent] = 0;
    std::memcpy(hd + o_off, off.data(), (static_cast<size_t>(Fc) + 1) * 4);
    char *dd = static_cast<char *>(d_sdict_.ensure((dict_bytes + 3) & ~static_cast<size_t>(3)));
    {
        void *hd_dev = nullptr;
        hip_check(hipHostGetDevicePointer(&hd_dev, hd, 0), "hipHostGetDevicePointer");
        kern::FetchSegments fs{};
        fs.n = 1; fs.dst[0] = dd; fs.src[0] = hd_dev; fs.words[0] = static_cast<uint32_t>((dict_bytes + 3) / 4);
        kern::fetch_segments(fs, s);
    }
    sdict_words_ = reinterpret_cast<const uint64_t *>(dd + o_words);
    sdict_hash_ = reinterpret_cast<const uint64_t *>(dd + o_hash);
    sdict_off_ = reinterpret_cast<const int32_t *>(dd + o_off);
    sdict_cls_ = reinterpret_cast<const int32_t *>(dd + o_cls);
    return true;
}

// A categorical cell that device_categorical_candidates recognised by (feature, 64-bit hash) alone: its 128 bytes, still in the pinned block
// the device published them to, are compared with the remembered ones HERE -- called while the device grows the tree, so the 260 KiB of
// PCIe-written lines are read off the critical path.  A difference means two categories share a 64-bit hash: the step is refused (nothing has
// joined the model yet) instead of continuing with the wrong category's name and candidate order.
void Engine::verify_pending_categories() {
    bool clash = false;
    for (const auto &pq : cat_pending_) clash = clash || std::memcmp(cat_items_[pq.first].name, pq.second, kCat) != 0;
    if (!cat_pending_.empty() && hooks::on(hooks::TEST_CAT_CLASH)) clash = true;   // test hook: pretend a remembered cell's bytes differ
    cat_pending_.clear();
    if (clash) cat_clash_ = true;
}

// ---- A3/A4: numeric split candidates ------------------------------------------------------------------------------------
// thresholds [F][B] of the rows in dobs (keys already transposed into d_kt): fixed ones (fit()), uniform (min/max + fma), or
// exact quantiles (radix multi-select; sample-splitter selection and 32-pass bisection kept as cross-checks / fallbacks).
// On return d_thr / d_thrkeys hold them on the device (the caller copies them to the host when it needs them there).
// The quantile target ranks depend on (global row count, n_bins) only: uploaded when they change (an RL loop calls step() with the same
// batch size over and over; the upload from pageable memory costs ~70 us of host time per call), through pinned memory.
int64_t *Engine::quantile_cum_device(const std::vector<int64_t> &cum, long long n_global, int B) {
    hipStream_t s = stream_;
    const bool grown = d_cum_.capacity() < sizeof(int64_t) * static_cast<size_t>(B);
    int64_t *d_cum = static_cast<int64_t *>(d_cum_.ensure(sizeof(int64_t) * B));
    if (grown || cum_cache_n_ != n_global || cum_cache_b_ != B) {
        int64_t *h = static_cast<int64_t *>(pin_cum_.ensure(sizeof(int64_t) * B));
        std::memcpy(h, cum.data(), sizeof(int64_t) * B);
        hip_check(hipMemcpyAsync(d_cum, h, sizeof(int64_t) * B, hipMemcpyHostToDevice, s), "H2D cum");
        hip_check(hipStreamSynchronize(s), "sync cum");   // the pinned block may be rewritten by the next call
        cum_cache_n_ = n_global;
        cum_cache_b_ = B;
    }
    return d_cum;
}
void Engine::numeric_thresholds(const float *dobs, int N, int F, int B, long long n_global, const uint32_t *d_kt, float *d_thr,
                                uint32_t *d_thrkeys, int pass1_chunks, uint16_t *d_codes_out, bool *codes_written) {
    if (codes_written) *codes_written = false;
    hipStream_t s = stream_;
    const gbrl_hip_metadata &md = model.meta;
    uint32_t *d_qflags = static_cast<uint32_t *>(d_qflags_.ensure(sizeof(uint32_t) * 4));  // [0,1] allocator, [2] overflow
    bool fast_quantile = false;
    // The target ranks depend on (global row count, n_bins) only: uploaded when they change (an RL loop calls step() with the same
    // batch size over and over; the upload from pageable memory costs ~70 us of host time per call), through pinned memory.
    auto upload_cum = [&](const std::vector<int64_t> &cum) -> int64_t * { return quantile_cum_device(cum, n_global, B); };
    auto bisection_quantiles = [&](const std::vector<int64_t> &cum) {
        // exact but slow: 32 counting passes (also the multi-GPU path: only integer counts cross ranks)
        int64_t *d_cum = upload_cum(cum);
        uint32_t *d_prefix = static_cast<uint32_t *>(d_prefix_.ensure(sizeof(uint32_t) * F * B));
        uint32_t *d_trial = static_cast<uint32_t *>(d_trial_.ensure(sizeof(uint32_t) * F * B));
        int64_t *d_counts = static_cast<int64_t *>(d_counts_.ensure(sizeof(int64_t) * F * (B + 1)));
        kern::qsel_init(d_prefix, d_trial, F, B, s);
        for (int bit = 31; bit >= 0; --bit) {
            hip_check(hipMemsetAsync(d_counts, 0, sizeof(int64_t) * F * (B + 1), s), "memset");
            kern::bin_rows(dobs, N, F, d_trial, B, /*strict=*/false, d_counts, nullptr, 0, 0, s);
            if (has_coll_) exchange(Red::SumI64, d_counts, static_cast<size_t>(F) * (B + 1));
            kern::qsel_update(d_prefix, d_trial, d_counts, d_cum, F, B, bit, bit - 1, s);
        }
        hip_check(hipMemcpyAsync(d_thrkeys, d_trial, sizeof(uint32_t) * F * B, hipMemcpyDeviceToDevice, s), "D2D keys");
    };
    std::vector<int64_t> cum;
    if (F > 0 && !fixed_thr_.empty()) {
        // fit(): the candidates were generated from the whole data set (fitter.cpp:134-150); this batch only bins against them
        if (fixed_thr_.size() != static_cast<size_t>(F) * B) throw HipError("internal: fixed thresholds do not match this model");
        hip_check(hipMemcpyAsync(d_thr, fixed_thr_.data(), sizeof(float) * fixed_thr_.size(), hipMemcpyHostToDevice, s), "H2D thresholds");
        kern::floats_to_keys(d_thr, d_thrkeys, fixed_thr_.size(), s);
    } else if (F > 0) {
        if (md.generator_type == GBRL_HIP_GEN_UNIFORM) {
            uint32_t *d_mm = static_cast<uint32_t *>(d_minmax_.ensure(sizeof(uint32_t) * 2 * F));
            {
                kern::FillSegments fz{};
                fz.n = 2;
                fz.dst[0] = d_mm; fz.words[0] = static_cast<uint32_t>(F); fz.value[0] = 0xffffffffu;
                fz.dst[1] = d_mm + F; fz.words[1] = static_cast<uint32_t>(F); fz.value[1] = 0u;
                kern::fill_segments(fz, s);
            }
            kern::column_minmax(d_kt, N, F, d_mm, d_mm + F, s);
            if (has_coll_) {
                // exchange as floats (max / min are exact)
                float *tmp = static_cast<float *>(d_trial_.ensure(sizeof(float) * 2 * F));
                kern::keys_to_floats(d_mm, tmp, 2 * static_cast<size_t>(F), s);
                kern::negate_f32(tmp, F, s);                 // min = -max(-x): minima and maxima in ONE max all-reduce
                exchange(Red::MaxF32, tmp, 2 * static_cast<size_t>(F));
                kern::negate_f32(tmp, F, s);
                kern::floats_to_keys(tmp, d_mm, 2 * static_cast<size_t>(F), s);
            }
            kern::uniform_thresholds(d_mm, d_mm + F, F, B, d_thr, s, d_thrkeys);
        } else {
            // split_candidate_generator.cpp:216-249: n_bins+1 equal-count buckets, threshold i = value at rank cum_i - 1.  With
            // fewer rows than buckets the remainder loop still gives the first n_samples buckets one row each, so cum_i =
            // min(i + 1, n_samples) >= 1: the ranks repeat at the column maximum (the reference grows valid trees there).
            cum = quantile_target_ranks(n_global, B);
            bool floats_done = false;
            // sharded fast path needs a power-of-two world (union sample of world*4096 keys sorted in LDS)
            const bool coll_fast = has_coll_ && (coll_.world_size & (coll_.world_size - 1)) == 0 && coll_.world_size <= 8;
            const bool radix_ok = !force_sample_select_ && B <= kern::radix_max_targets() && n_global < (1ll << 32);
            if (force_bisection_ || (has_coll_ && !coll_fast && !radix_ok)) {
                bisection_quantiles(cum);
            } else if (!has_coll_ && !force_sample_select_ && !force_radix_ && kern::sort_quantiles_fits(N, B)) {
                // RL-sized batch: the column fits in LDS -- sort it and read the ranks (one launch)
                int64_t *d_cum = upload_cum(cum);
                // (the sort kernel also writes the class codes of its feature: no separate binning launch)
                const bool no_fuse = [] { const char *e = hooks::raw(hooks::SORT_NO_CODES); return e && e[0] == '1'; }();   /* read per call: the tests flip it */   // test / measurement hook
                uint16_t *cdst = no_fuse ? nullptr : d_codes_out;
                kern::sort_quantiles(d_kt, N, F, d_cum, B, d_thrkeys, d_thr, s, cdst);
                if (cdst && codes_written) *codes_written = true;
                floats_done = true;
                last_quantile_fallback_ = false;
            } else if (!force_sample_select_ && B <= kern::radix_max_targets() && n_global < (1ll << 32)) {
                // exact MSD radix multi-select, four counting passes over the transposed keys (radix_select.hip).  Row-sharded
                // runs sum the digit counts of every pass over ranks (any world size): 4 all-reduces per step.
                int64_t *d_cum = upload_cum(cum);
                void *d_rs = d_radix_state_.ensure(kern::radix_state_bytes(F, B));
                uint32_t *d_rp = static_cast<uint32_t *>(d_radix_partial_.ensure(kern::radix_partial_bytes(F)));
                uint32_t *d_rl = static_cast<uint32_t *>(d_qlists_.ensure(kern::radix_list_bytes(N, F)));
                kern::RadixComm comm{};
                if (has_coll_) {
                    comm.ctx = this;
                    comm.allreduce_sum_i64 = &Engine::radix_exchange_trampoline;
                    comm.stream_ordered = rccl_comm_ != nullptr;
                    comm.gbuf = static_cast<int64_t *>(d_counts_.ensure(sizeof(int64_t) * kern::radix_exchange_words(F)));
                    comm.partial_global = static_cast<uint32_t *>(d_radix_global_.ensure(kern::radix_global_partial_bytes(F)));
                }
                // (one GPU: the selection also reports #{keys <= threshold}, from which the ROOT's class counts follow -- grow_tree, root_le)
                uint32_t *d_le = has_coll_ ? nullptr : static_cast<uint32_t *>(d_root_le_.ensure(sizeof(uint32_t) * (static_cast<size_t>(F) * B + F)));
                const int rc = kern::radix_select(d_kt, N, F, d_cum, B, d_rs, d_rp, d_rl, d_thrkeys, s, has_coll_ ? &comm : nullptr, pass1_chunks, d_le);
                root_le_ = d_le;
                if (rc != 0) throw HipError(rc == 2 ? "allreduce failed" : "radix select failed");
                last_quantile_fallback_ = false;
            } else {
                fast_quantile = true;
                kern::QuantilePlan plan = kern::quantile_plan(N);
                if (has_coll_) { plan.sample = 4096; plan.n_split = kern::kQuantileMaxSplit; }   // identical on every rank
                // budget of the extracted class lists: a quarter of the data is ample when the targets are few against the classes
                // (<= 256 targets, 1024 classes); with more targets nearly every class holds one, so the lists can be the whole data
                const size_t all_keys = static_cast<size_t>(N) * F;
                const uint32_t max_elems = static_cast<uint32_t>(B > 256 ? all_keys : std::min<size_t>(all_keys, std::max<size_t>(1u << 20, all_keys / 4)));
                int64_t *d_cum = upload_cum(cum);
                uint32_t *d_split = static_cast<uint32_t *>(d_splitters_.ensure(sizeof(uint32_t) * 2 * static_cast<size_t>(F) * kern::kQuantileMaxSplit));
                uint32_t *d_split_bfs = d_split + static_cast<size_t>(F) * kern::kQuantileMaxSplit;
                uint32_t *d_cc = static_cast<uint32_t *>(d_ccounts_.ensure(sizeof(uint32_t) * static_cast<size_t>(plan.n_chunks) * F * kern::kQuantileClasses));
                uint32_t *d_coff = static_cast<uint32_t *>(d_c2l_.ensure(sizeof(uint32_t) * F * kern::kQuantileClasses));
                uint32_t *d_toff = static_cast<uint32_t *>(d_tgt_list_.ensure(sizeof(uint32_t) * 2 * static_cast<size_t>(F) * B));
                uint32_t *d_tlen = d_toff + static_cast<size_t>(F) * B;
                uint32_t *d_tr = static_cast<uint32_t *>(d_tgt_rank_.ensure(sizeof(uint32_t) * F * B));
                uint32_t *d_lists = static_cast<uint32_t *>(d_qlists_.ensure(sizeof(uint32_t) * max_elems));
                hip_check(hipMemsetAsync(d_coff, 0xff, sizeof(uint32_t) * F * kern::kQuantileClasses, s), "memset");
                hip_check(hipMemsetAsync(d_qflags, 0, sizeof(uint32_t) * 4, s), "memset");
                int64_t *d_gcounts = nullptr;
                if (has_coll_) {
                    // every rank contributes a 4096-key sample per feature; the union is sorted identically everywhere
                    const int S = 4096, SU = S * coll_.world_size;
                    uint32_t *d_samp = static_cast<uint32_t *>(d_prefix_.ensure(sizeof(uint32_t) * static_cast<size_t>(F) * S));
                    int64_t *d_uni = static_cast<int64_t *>(d_counts_.ensure(sizeof(int64_t) * std::max<size_t>(static_cast<size_t>(F) * SU, static_cast<size_t>(F) * kern::kQuantileClasses)));
                    kern::sample_only(d_kt, N, F, S, d_samp, s);
                    hip_check(hipMemsetAsync(d_uni, 0, sizeof(int64_t) * static_cast<size_t>(F) * SU, s), "memset");
                    kern::place_sample(d_samp, F, S, coll_.rank, SU, d_uni, s);
                    exchange(Red::SumI64, d_uni, static_cast<size_t>(F) * SU);
                    kern::union_splitters(d_uni, F, SU, plan.n_split, d_split, d_split_bfs, s);
                    kern::class_count(d_kt, N, F, plan, d_split_bfs, d_cc, s);
                    d_gcounts = d_uni;   // reuse (the union sample is consumed)
                    kern::counts_to_i64(d_cc, plan.n_chunks, static_cast<size_t>(F) * kern::kQuantileClasses, d_gcounts, s);
                    exchange(Red::SumI64, d_gcounts, static_cast<size_t>(F) * kern::kQuantileClasses);
                } else {
                    kern::sample_splitters(d_kt, N, F, plan, d_split, d_split_bfs, s);
                    hip_check(hipGetLastError(), "sample_splitters launch");
                    kern::class_count(d_kt, N, F, plan, d_split_bfs, d_cc, s);
                    hip_check(hipGetLastError(), "class_count launch");
                }
                kern::quantile_targets(d_cc, d_gcounts, d_split, d_cum, F, B, plan, d_coff, d_toff, d_tlen, d_tr, d_thrkeys, d_qflags, max_elems,
                                       d_qflags + 2, s);
                kern::quantile_extract(d_kt, N, F, plan, d_split_bfs, d_coff, d_cc, d_lists, s);
                hip_check(hipGetLastError(), "quantile_extract launch");
                if (has_coll_) {
                    // the lists stay on their ranks; the order statistic of their union is found by 32 counting rounds
                    uint32_t *d_pref = static_cast<uint32_t *>(d_trial_.ensure(sizeof(uint32_t) * static_cast<size_t>(F) * B));
                    int64_t *d_scnt = static_cast<int64_t *>(d_selcnt_.ensure(sizeof(int64_t) * (static_cast<size_t>(F) * B + 1)));
                    hip_check(hipMemsetAsync(d_pref, 0, sizeof(uint32_t) * static_cast<size_t>(F) * B, s), "memset");
                    for (int bit = 31; bit >= 0; --bit) {
                        kern::select_count(d_lists, d_toff, d_tlen, d_pref, bit, F * B, d_scnt, s);
                        hip_check(hipGetLastError(), "select_count launch");
                        exchange(Red::SumI64, d_scnt, static_cast<size_t>(F) * B);
                        kern::select_update(d_pref, d_scnt, d_toff, d_tr, bit, F * B, d_thrkeys, s);
                        hip_check(hipGetLastError(), "select_update launch");
                    }
                } else {
                    kern::quantile_select(d_lists, d_toff, d_tlen, d_tr, F * B, d_thrkeys, s);
                    hip_check(hipGetLastError(), "quantile_select launch");
                }
            }
            if (!floats_done) kern::keys_to_floats(d_thrkeys, d_thr, static_cast<size_t>(F) * B, s);   // (the LDS sort writes the floats itself)
        }
        uint32_t qflags[4] = {0, 0, 0, 0};
        if (fast_quantile) {
            hip_check(hipMemcpyAsync(qflags, d_qflags, sizeof(qflags), hipMemcpyDeviceToHost, s), "D2H flags");
            hip_check(hipStreamSynchronize(s), "sync");
            if (has_coll_) {   // the fallback decision must be the same on every rank
                int64_t *d_flag = static_cast<int64_t *>(d_selcnt_.ensure(sizeof(int64_t) * 2));
                int64_t hv = qflags[2];
                hip_check(hipMemcpyAsync(d_flag, &hv, sizeof(hv), hipMemcpyHostToDevice, s), "H2D flag");
                exchange(Red::SumI64, d_flag, 1);
                hip_check(hipMemcpyAsync(&hv, d_flag, sizeof(hv), hipMemcpyDeviceToHost, s), "D2H flag");
                hip_check(hipStreamSynchronize(s), "sync");
                qflags[2] = hv != 0;
            }
            if (qflags[2] != 0) {  // a class list outgrew its budget (pathological value distribution): redo exactly, slowly
                bisection_quantiles(cum);
                kern::keys_to_floats(d_thrkeys, d_thr, static_cast<size_t>(F) * B, s);
                last_quantile_fallback_ = true;
            } else {
                last_quantile_fallback_ = false;
            }
        }
    }
}

}  // namespace gbrl
