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
#include "engine.h"
#include "hooks.h"

#include <numeric>
#include <random>
#include "cat_hash.h"
#include "hash_order_replay.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <unordered_map>

namespace gbrl {

using kern::Chunk;
using kern::FeatureSlot;
using kern::NodeSplit;


namespace detail {

struct HCond {       // splitCondition (types.h:64-70) + what the kernels need
    int fslot;       // feature slot (numeric f, or F + categorical c)
    int feat_idx;    // feature index as stored in the model (within its numeric / categorical block)
    float value;     // numeric threshold, +inf for categorical (split_candidate_generator.cpp:155)
    int bin;         // numeric: threshold index; categorical: class id
    bool is_cat;
    bool dir;
    float edge_w;
    int cat_cand;    // index into cat candidate strings, -1 for numeric
};

struct HNode {
    int depth = 0;
    int seg_start = 0;
    int n_local = 0;        // rows of this rank in the node
    long long n_global = 0;  // rows over all ranks
    std::vector<HCond> path;
    int left = -1, right = -1;
    int parent = -1;
    int hist_slot = -1;     // slot of this node's histogram in its level's buffer
    bool leaf = false;
};

// Everything Engine::grow_tree needs from the preparation stages of step().
struct GrowCtx {
    int N, F, Fc, D, B, MD, NB, FG, Fp, n_groups, n_slots, n_cand, chunk_rows;
    long long n_global;
    bool cosine, oblivious;
    const std::vector<kern::FeatureSlot> *slots;
    const std::vector<float> *cand_w;
    const std::vector<int32_t> *cand_ref;
    const std::vector<int> *ref_to_internal;
    const std::vector<int32_t> *cand_slot;
    bool const_cacheable;               // numeric-only step: the constants above live in Engine::step_const_
    const std::vector<CatCandidate> *cat_cands;
    bool prefix_cacheable;     // mixed step: numeric table prefixes stay on the device, categorical tails are uploaded per step
    int n_num_cand, cand_cap;  // numeric candidates (= prefix length), capacity of the fixed table layout
    char *pub_thr_dev, *pub_scales_dev; // device addresses of the pinned copies below (kern::publish_pair / the growth kernel write them)
    size_t pub_thr_bytes;
    const float *h_thr;                 // pinned; valid once the stream has passed the copy enqueued behind the binning
    const kern::StepScales *h_scales;   // pinned, same
    const float *d_thr;
    const uint32_t *d_thrkeys;   // [F][B] ordered keys of the thresholds
    const uint32_t *root_le;     // [F][B] #{keys <= threshold} from the radix selection (one GPU, numeric-only steps), else null
    const uint32_t *d_kt;        // [F][N] feature-major ordered keys of the observations (null when F == 0)
    const uint16_t *d_codes;
    const uint16_t *d_codes_fm;  // [F][N] feature-major copy of the numeric codes (fused preparation only), else null
    const int32_t *d_qg;
    const float *dgrads;
    const float *d_meanden;      // L2: [D] mean | [D] std + 1e-8f of the build gradients' standardisation; null for Cosine (raw gradients)
    kern::StepScales *d_scales;
};

}  // namespace detail

using detail::CatCandidate;
using detail::GrowCtx;
using detail::HCond;
using detail::HNode;

namespace {

// Packs many small host arrays into one pinned block and uploads them with ONE async copy; put() returns the DEVICE
// address the array will have.  The pinned block must not be refilled before the copy has executed (the caller's
// per-level synchronisation guarantees it).
class Stager {
   public:
    Stager(PinnedBuf &pin, DevBuf &dev, size_t cap, hipStream_t s) : s_(s) {
        host_ = static_cast<char *>(pin.ensure(cap));
        dev_ = static_cast<char *>(dev.ensure(cap));
        cap_ = cap;
    }
    void reset() { used_ = 0; }
    template <typename T>
    T *put(const T *src, size_t n) {
        const size_t bytes = n * sizeof(T);
        if (used_ + bytes + 256 > cap_) throw HipError("internal: staging buffer overflow");
        if (bytes) std::memcpy(host_ + used_, src, bytes);
        T *d = reinterpret_cast<T *>(dev_ + used_);
        used_ += (bytes + 255) & ~static_cast<size_t>(255);
        return d;
    }
    template <typename T>
    T *reserve(size_t n) {     // the device address put() would return, without touching the host copy (the block is already uploaded)
        const size_t bytes = n * sizeof(T);
        if (used_ + bytes + 256 > cap_) throw HipError("internal: staging buffer overflow");
        T *d = reinterpret_cast<T *>(dev_ + used_);
        used_ += (bytes + 255) & ~static_cast<size_t>(255);
        return d;
    }
    void flush() {
        if (used_) hip_check(hipMemcpyAsync(dev_, host_, used_, hipMemcpyHostToDevice, s_), "H2D staged descriptors");
    }
    const void *device_base() const { return dev_; }
    char *host_base() const { return host_; }

   private:
    hipStream_t s_;
    char *host_ = nullptr, *dev_ = nullptr;
    size_t cap_ = 0, used_ = 0;
};


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

}  // namespace

// ---- A5 on the device ------------------------------------------------------------------------------------------------------
// Finds the distinct (feature, cell) pairs of the batch and their first rows with per-feature hash tables on the device, hands
// the few distinct cells to the host, which inserts them into the SAME container in the SAME order as the reference's scan
// (feature-major, first occurrence; split_candidate_generator.cpp:119-129) -- so the candidate order (Q8) is the reference's --
// and uploads the candidate dictionary for k_cat_step_codes.  Returns false (caller falls back to the host scan) when the
// batch has more distinct categories than the Fc*B the reference keeps (it then ranks them by mean gradient norm), a table
// overflowed, or two different cells collided on their 64-bit hash.
// ---- A5, row-sharded, more distinct categories than Fc * n_bins: the reference keeps the Fc * n_bins categories with the largest mean
// gradient norm (split_candidate_generator.cpp:141-149).  Its per-category total is a float32 sum in ROW order and its candidate order
// is the iteration order of its hash map followed by std::sort -- both reproduced here exactly: (1) every rank scans its own cells on
// the host in the reference's loop order (feature-major, row-minor); (2) the ranks' distinct (feature, category) pairs are all-gathered
// in rank order, which is global first-occurrence order; (3) the totals are accumulated by ONE rank at a time in rank order, each
// starting from the running totals of the ranks before it (P small broadcasts), so every addition happens in global row order;
// (4) every rank builds the reference's container from the global list and ranks it.  Slow (host scan, P rounds) and rare.
// GBRL_HIP_DEVICE_LEVELS=1 (opt-in device-planned level loop), latched at the first use: the cached root row list and the choice of
// the loop must see the same value for the whole process.
static bool device_levels_requested() {
    const bool v = [] { const char *e = hooks::raw(hooks::DEVICE_LEVELS); return e && e[0] == '1'; }();
    return v;
}
// Host side of the copy-free hand-overs: poll a sequence word in coherent pinned memory; every 16384 polls ask the stream for errors
// (a faulted kernel never publishes) and give up after GBRL_HIP_SPIN_SECONDS (default 120) of wall clock -- a hung kernel must not
// spin a core forever, and a slow but healthy run (counter profiling, several ranks sharing one device) must not be declared dead.
// Before giving up the stream is synchronised: kernels still in flight would otherwise keep storing into the pinned result blocks and
// pools that the next call reuses; a stream that does drain turns the timeout into an ordinary completion.
static void spin_until_published(volatile uint32_t *flag, uint32_t seq, hipStream_t s, const char *what) {
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
// split_candidate_generator.cpp:216-249: n_bins+1 equal-count buckets, threshold i = value at rank cum_i - 1.  With fewer rows than buckets
// the remainder loop still gives the first n_samples buckets one row each, so cum_i = min(i + 1, n_samples) >= 1: the ranks repeat at the
// column maximum (the reference grows valid trees there).
static std::vector<int64_t> quantile_target_ranks(long long n_global, int B) {
    std::vector<int64_t> cum(B);
    const long long per = n_global / (B + 1), rem = n_global % (B + 1);
    long long run = 0;
    for (int i = 0; i < B; ++i) { run += per + (i < rem ? 1 : 0); cum[i] = run; }
    return cum;
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
        if (has_coll_ && oblivious) kern::fill_f32(d_scores, static_cast<size_t>(n_act) * n_cand, -INFINITY, s);
        // last level on one GPU: the derived siblings are scored but not written back (nothing subtracts from them any more)
        const bool skip_hook = [] { const char *e = hooks::raw(hooks::KEEP_LAST_DERIVED); return e && e[0] == '1'; }();   // measurement hook
        const bool drop_derived = !has_coll_ && !skip_hook && depth > 0 && depth == MD - 1;
        if (own_slots > 0)
            kern::score_candidates(d_hist, d_hist_prev, depth > 0 ? d_sub_par : nullptr, d_sub_sib, n_act, Fp, NB, D, d_slots, own_slots, d_thr, B, n_cand, md.min_data_in_leaf, cosine ? 1 : 0,
                                   d_scales, d_path_len, d_path_slot, d_path_val, d_path_bin, d_scores, d_parent, d_cand_w, d_cand_ref, d_isroot,
                                   oblivious ? nullptr : d_am_v, d_am_i, s, has_coll_ ? coll_lo : 0, !drop_derived, oblivious ? nullptr : d_am_s, oblivious ? nullptr : d_am_n);
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
        kern::resolve_splits(d_am_v, d_am_i, oblivious ? am_parts : own_slots, d_best_idx, d_best_score, oblivious, n_act, d_ref_to_internal, d_cand_slot, d_slots, d_hist, nullptr, Fp, NB, D, d_resolved,
                             d_counts4, max_front, d_seg_starts, d_cursors, c.d_thrkeys, B, s, publish_in_resolve ? h_res_dev : nullptr, d_flag, seq, d_pub_done,
                             drop_derived ? d_hist_prev : nullptr, drop_derived ? d_sub_par : nullptr, drop_derived ? d_sub_sib : nullptr, near_level ? &near_detect : nullptr);
        if (has_coll_) {
            // the level's winner over all ranks: every rank holds the best of ITS features and the child sizes it induces
            const int n_win = oblivious ? 1 : n_act;
            const size_t gwords = static_cast<size_t>(coll_P) * (n_win + 2 * n_act);
            kern::winner_pack(d_best_idx, d_best_score, d_counts4, max_front, n_win, n_act, coll_.rank, d_gather, s, coll_P);
            exchange(Red::SumI64, d_gather, gwords);
            kern::winner_adopt(d_gather, coll_P, n_win, n_act, oblivious, d_ref_to_internal, d_cand_slot, d_slots, d_seg_starts, c.d_thrkeys, B, d_best_idx, d_best_score,
                               d_counts4, max_front, d_resolved, d_cursors, s);
            int64_t *d_right_local = d_counts4 + 2 * static_cast<size_t>(max_front);   // (cleared by winner_adopt)
            if (!count_chunks.empty())
                kern::count_right(d_rows[cur], d_codes, c.d_kt, N, d_count_chunks, static_cast<int>(count_chunks.size()), d_resolved, d_right_local, s);
            // global left sizes -> this rank's, and the completed result block to the host: one launch
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

    // global row count (rows are sharded over ranks)
    long long n_global = N;
    if (has_coll_) {
        int64_t *tmp = static_cast<int64_t *>(d_ntotal_.ensure(sizeof(int64_t)));
        int64_t hv = N;
        hip_check(hipMemcpyAsync(tmp, &hv, sizeof(hv), hipMemcpyHostToDevice, s), "H2D n");
        exchange(Red::SumI64, tmp, 1);
        hip_check(hipMemcpyAsync(&hv, tmp, sizeof(hv), hipMemcpyDeviceToHost, s), "D2H n");
        hip_check(hipStreamSynchronize(s), "sync");
        n_global = hv;
    }

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
    const int chunk_rows = static_cast<int>(std::min<long long>(65536, std::max<long long>(4096, 2 * ((n_global + 31) / 32))));
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
        double *d_smsg = has_coll_ ? static_cast<double *>(d_maxbits_.ensure(sizeof(double) * static_cast<size_t>(D) * (world + 1))) : nullptr;
        auto exchange_stats = [&](double *st) {
            if (!has_coll_) return;
            kern::stats_pack(st, D, world, coll_.rank, d_smsg, s);
            exchange(Red::SumF64, d_smsg, static_cast<size_t>(D) * (world + 1));
            kern::stats_unpack(d_smsg, D, world, st, s);
        };
        // RL-sized batch on one GPU: statistics and quantisation in ONE launch with the same reduction tree (kern::small_stats)
        if (!has_coll_ && !no_small_stats && n_global == N && kern::small_stats(dgrads, N, D, !cosine, chunk_rows, d_stat, d_meanden, d_scales, d_qg, s)) {
            stats_fused = true;
            if (!cosine) { d_mean = d_meanden; d_den = d_meanden + D; }
        } else {
        kern::column_sums(dgrads, N, D, nullptr, d_part, nblk, d_stat, s);
        exchange_stats(d_stat);
        if (!cosine) {
            kern::stats_mean(d_stat, n_global, D, d_meanden, s);
            kern::column_sums(dgrads, N, D, d_meanden, d_part, nblk, d_stat2, s);
            exchange_stats(d_stat2);
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
