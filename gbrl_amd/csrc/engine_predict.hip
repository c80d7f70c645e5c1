// engine_predict.hip -- the device mirror of the ensemble (append-only structure-of-arrays + the packed records the fast predict
// kernels read) and GBRL::predict (see engine.h).
#include "engine.h"
#include "hooks.h"

#include <functional>
#include "cat_hash.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <unordered_map>

namespace gbrl {

constexpr int kChainMinTrees = 512;   // below: the tiled kernels are as fast (measured, scripts/predict_latency.py)
constexpr int kChainMaxRows = 8192;   // 5000 rows x 20 000 trees: 1.58 -> 0.88 ms, 8192 x 5000: 0.39 -> 0.32 ms; beyond, the leaf search alone costs as much

// ===================================================================================================== predict
void Engine::sync_model_to_device() {
    hipStream_t s = stream_;
    const gbrl_hip_metadata &md = model.meta;
    const size_t T = md.n_trees, L = md.n_leaves, S = model.split_rows(), MD = md.max_depth, D = md.output_dim;
    if (mirror_version_ == model.version) return;
    // dictionary ids for the categorical conditions (strings are compared on the host once; the device compares ids)
    if (up_splits_ > S || up_trees_ > T || up_leaves_ > L) { up_splits_ = up_trees_ = up_leaves_ = 0; grd_up_nodes_ = 0; cat_dict_.clear(); cat_ids_host_.clear(); cond_pack_host_.clear(); dict_version_ = static_cast<size_t>(-1); }
    cat_ids_host_.resize(S * MD, 0);
    for (size_t c = up_splits_ * MD; c < S * MD; ++c) {
        if (model.is_numerics[c]) continue;
        const int f = model.feature_indices[c];
        std::string name(&model.categorical_values[c * kCat], kCat);
        int id = 0;
        for (size_t z = 0; z < cat_dict_.size(); ++z)
            if (cat_dict_[z].first == f && cat_dict_[z].second == name) { id = static_cast<int>(z) + 1; break; }
        if (id == 0) { cat_dict_.emplace_back(f, name); id = static_cast<int>(cat_dict_.size()); }
        cat_ids_host_[c] = id;
    }
    // The appended slices (one new tree after a step: a dozen pieces of a few hundred bytes) are collected here and leave together at the end
    // (flush_segments): staged in ONE pinned block and copied by one kernel launch, instead of a hipMemcpyAsync from pageable memory per
    // piece plus a stream synchronisation -- 0.1 ms per predict-after-step in an RL loop (round 4).
    struct Seg { char *dst; const char *src; size_t bytes; };
    std::vector<Seg> segs;
    auto append = [&](DevBuf &buf, const void *host, size_t elem, size_t old_n, size_t new_n) {
        char *p = static_cast<char *>(buf.ensure_keep(std::max<size_t>(new_n, 1) * elem, old_n * elem, s));
        if (new_n > old_n) segs.push_back({p + old_n * elem, static_cast<const char *>(host) + old_n * elem, (new_n - old_n) * elem});
    };
    // records followed by `pad` zero elements that the kernels read past the last tree: when the buffer did not move, the old padding is still
    // in place and only the new records and the padding's extension travel (the padding is 128 KiB for the value records)
    auto append_padded = [&](DevBuf &buf, const float *host, size_t old_recs, size_t new_recs, size_t rec, size_t pad) {
        const void *before = buf.raw();
        char *p = static_cast<char *>(buf.ensure_keep((new_recs * rec + pad) * 4, old_recs * rec * 4, s));
        if (p != before || old_recs == 0 || old_recs > new_recs) {
            segs.push_back({p + old_recs * rec * 4, reinterpret_cast<const char *>(host + old_recs * rec), ((new_recs - old_recs) * rec + pad) * 4});
        } else if (new_recs > old_recs) {
            segs.push_back({p + old_recs * rec * 4, reinterpret_cast<const char *>(host + old_recs * rec), (new_recs - old_recs) * rec * 4});
            segs.push_back({p + (old_recs * rec + pad) * 4, reinterpret_cast<const char *>(host + old_recs * rec + pad), (new_recs - old_recs) * rec * 4});
        }
    };
    auto flush_segments = [&]() {
        size_t total = 0;
        for (const Seg &g : segs) total += (g.bytes + 15) & ~static_cast<size_t>(15);
        if (segs.empty()) return;
        if (segs.size() <= 32 && total <= (size_t(1) << 20)) {
            hip_check(hipEventSynchronize(ev_model_stage_), "sync");   // the previous use of the staging block has been read (normally long ago)
            char *st = static_cast<char *>(pin_model_stage_.ensure(total));
            void *st_dev = nullptr;
            hip_check(hipHostGetDevicePointer(&st_dev, st, 0), "hipHostGetDevicePointer");
            size_t off = 0;
            for (size_t i = 0; i < segs.size(); i += kern::kStageSegments) {
                kern::StageSegments ss{};
                for (size_t k = i; k < segs.size() && k < i + kern::kStageSegments; ++k) {
                    std::memcpy(st + off, segs[k].src, segs[k].bytes);
                    ss.dst[ss.n] = segs[k].dst; ss.src_off[ss.n] = static_cast<uint32_t>(off); ss.bytes[ss.n] = static_cast<uint32_t>(segs[k].bytes);
                    ++ss.n;
                    off += (segs[k].bytes + 15) & ~static_cast<size_t>(15);
                }
                kern::stage_copy(ss, st_dev, s);
            }
            hip_check(hipEventRecord(ev_model_stage_, s), "hipEventRecord");
        } else {   // a whole loaded ensemble: ordinary copies, and the host arrays must outlive them
            for (const Seg &g : segs) hip_check(hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyHostToDevice, s), "H2D model");
            hip_check(hipStreamSynchronize(s), "sync model upload");
        }
        segs.clear();
    };
    append(m_tree_indices_, model.tree_indices.data(), 4, up_trees_, T);
    append(m_depths_, model.depths.data(), 4, up_splits_, S);
    append(m_feature_indices_, model.feature_indices.data(), 4, up_splits_ * MD, S * MD);
    append(m_feature_values_, model.feature_values.data(), 4, up_splits_ * MD, S * MD);
    append(m_is_numerics_, model.is_numerics.data(), 1, up_splits_ * MD, S * MD);
    append(m_cat_ids_, cat_ids_host_.data(), 4, up_splits_ * MD, S * MD);
    // packed (feature | ~categorical feature, threshold bits | category id) pairs per split row: read through the scalar cache
    // by the fast oblivious predict kernel
    cond_pack_host_.resize(S * MD * 2, 0);
    for (size_t c = up_splits_ * MD; c < S * MD; ++c) {
        const bool num = model.is_numerics[c] != 0;
        int32_t tv;
        std::memcpy(&tv, &model.feature_values[c], sizeof(tv));
        cond_pack_host_[2 * c] = num ? model.feature_indices[c] : ~model.feature_indices[c];
        cond_pack_host_[2 * c + 1] = num ? tv : cat_ids_host_[c];
    }
    append(m_cond_pack_, cond_pack_host_.data(), 4, up_splits_ * MD * 2, S * MD * 2);
    append(m_values_, model.values.data(), 4, up_leaves_ * D, L * D);
    append(m_ineq_, model.inequality_directions.data(), 1, up_leaves_ * MD, L * MD);
    // Second-generation oblivious kernel (predict_obl2.hip): per tree a right-aligned condition record and the leaf values
    // pre-swizzled as [worker][leaf][DMAX/4]
    if (model.oblivious() && kern::obl2_feasible(static_cast<int>(MD), static_cast<int>(D), false)) {
        const size_t MX = kern::obl2_levels(static_cast<int>(MD)), DMAX = kern::obl2_padded_outputs(static_cast<int>(D)), DW = DMAX / 4;
        const size_t LS = size_t(1) << MX, VT = LS * DMAX;   // leaves padded to 2^levels: the kernel's tree stride is a compile-time constant
        constexpr size_t kPadTrees = 32;                     // the kernels fetch whole groups of records (<= 32 trees) past the last tree
        if (up_trees_ == 0) { cond_ra_host_.clear(); values_sw_host_.clear(); }
        cond_ra_host_.resize((T + kPadTrees) * 2 * MX);
        // the kernel loads whole groups (<= 16 trees) and up to 4 x 1024 x 16 bytes per block unconditionally: zero padding behind the last tree
        const size_t vpad = kPadTrees * VT + (size_t(64) << 10) / sizeof(float);
        values_sw_host_.resize(T * VT + vpad);
        std::fill(values_sw_host_.begin() + static_cast<long>(T * VT), values_sw_host_.end(), 0.0f);
        int32_t inf_bits;
        const float inf = std::numeric_limits<float>::infinity();
        std::memcpy(&inf_bits, &inf, sizeof(inf_bits));
        for (size_t t = up_trees_; t < T; ++t) {
            const size_t depth = static_cast<size_t>(model.depths[t]);
            int32_t *cr = &cond_ra_host_[t * 2 * MX];
            for (size_t d = 0; d < MX - depth; ++d) { cr[2 * d] = 0; cr[2 * d + 1] = inf_bits; }
            for (size_t d = 0; d < depth; ++d) {
                cr[2 * (MX - depth + d)] = cond_pack_host_[2 * (t * MD + d)];
                cr[2 * (MX - depth + d) + 1] = cond_pack_host_[2 * (t * MD + d) + 1];
            }
            float *vs = &values_sw_host_[t * VT];
            std::fill(vs, vs + VT, 0.0f);
            const size_t l0 = static_cast<size_t>(model.tree_indices[t]);
            for (size_t leaf = 0; leaf < (size_t(1) << depth); ++leaf)
                for (size_t j = 0; j < D; ++j) vs[((j / DW) * LS + leaf) * DW + (j % DW)] = model.values[(l0 + leaf) * D + j];
        }
        int32_t inf_pad[2] = {0, inf_bits};
        for (size_t t = T; t < T + kPadTrees; ++t)
            for (size_t d = 0; d < MX; ++d) { cond_ra_host_[(t * MX + d) * 2] = inf_pad[0]; cond_ra_host_[(t * MX + d) * 2 + 1] = inf_pad[1]; }
        append(m_cond_ra_, cond_ra_host_.data(), 4, up_trees_ * 2 * MX, (T + kPadTrees) * 2 * MX);
        append_padded(m_values_sw_, values_sw_host_.data(), up_trees_, T, VT, vpad);
    }
    // Greedy ensembles: rebuild every new tree as a binary tree from its leaves' paths (leaves are stored depth-first, left
    // first; fitter.cpp:364-365), for the descent of k_predict_grd.  A tree whose leaves do not form a proper binary tree (a
    // hand-edited model file), or a depth-0 tree (Q7), switches the fast path off for the whole ensemble.
    if (!model.oblivious()) {
        if (up_trees_ == 0) { grd_nodes_host_.clear(); grd_off_host_.assign(1, 0); grd_ok_ = true; grd_max_nodes_ = 0; grd_max_leaves_ = 1; }
        for (size_t t = up_trees_; t < T; ++t) {
            const int l0 = model.tree_indices[t], l1 = t + 1 < T ? model.tree_indices[t + 1] : static_cast<int>(L);
            const size_t base = grd_nodes_host_.size() / 4;
            bool ok = l1 > l0;
            // same condition (depth d) for two leaves?
            auto same_cond = [&](int a, int b, int d) {
                const size_t ca = static_cast<size_t>(a) * MD + d, cb = static_cast<size_t>(b) * MD + d;
                if (model.is_numerics[ca] != model.is_numerics[cb] || model.feature_indices[ca] != model.feature_indices[cb]) return false;
                if (model.is_numerics[ca]) return std::memcmp(&model.feature_values[ca], &model.feature_values[cb], 4) == 0;
                return cat_ids_host_[ca] == cat_ids_host_[cb];
            };
            // build(lo, hi, d): leaves [lo, hi) share their first d conditions; returns the child code
            std::function<int(int, int, int)> build = [&](int lo, int hi, int d) -> int {
                if (!ok) return -1;
                if (hi - lo == 1) {
                    if (model.depths[lo] != d) ok = false;
                    return ~(lo - l0);
                }
                if (d >= static_cast<int>(MD)) { ok = false; return -1; }
                int mid = lo;
                for (int q = lo; q < hi; ++q) {
                    if (model.depths[q] <= d || !same_cond(lo, q, d)) { ok = false; return -1; }
                    const bool right = model.inequality_directions[static_cast<size_t>(q) * MD + d] != 0;
                    if (!right) { if (q != mid) { ok = false; return -1; } ++mid; }   // left leaves first, contiguous
                }
                if (mid == lo || mid == hi) { ok = false; return -1; }
                const size_t me = grd_nodes_host_.size() / 4 - base;
                grd_nodes_host_.insert(grd_nodes_host_.end(), {0, 0, 0, 0});
                const size_t c = static_cast<size_t>(lo) * MD + d;
                int32_t tv;
                std::memcpy(&tv, &model.feature_values[c], sizeof(tv));
                const bool num = model.is_numerics[c] != 0;
                const int left = build(lo, mid, d + 1), right = build(mid, hi, d + 1);
                int32_t *nd = &grd_nodes_host_[(base + me) * 4];
                nd[0] = num ? model.feature_indices[c] : ~model.feature_indices[c];
                nd[1] = num ? tv : cat_ids_host_[c];
                nd[2] = left;
                nd[3] = right;
                return static_cast<int>(me);
            };
            if (ok && l1 - l0 == 1) ok = false;                 // depth-0 tree: its leaf never passes in the reference (Q7)
            if (ok) { const int root = build(l0, l1, 0); if (root != 0) ok = false; }
            if (!ok) { grd_ok_ = false; grd_nodes_host_.resize(base * 4); }
            grd_off_host_.push_back(static_cast<int32_t>(grd_nodes_host_.size() / 4));
            grd_max_nodes_ = std::max<int>(grd_max_nodes_, static_cast<int>(grd_nodes_host_.size() / 4 - base));
            grd_max_leaves_ = std::max(grd_max_leaves_, l1 - l0);
        }
        append(m_grd_nodes_, grd_nodes_host_.data(), 4, grd_up_nodes_ * 4, grd_nodes_host_.size());
        append(m_grd_off_, grd_off_host_.data(), 4, 0, grd_off_host_.size());
        grd_up_nodes_ = grd_nodes_host_.size() / 4;
        // Second-generation kernel, greedy mode (predict_obl2.hip): one record per tree = its leaf values pre-swizzled as
        // [worker][leaf < 2^levels][DMAX/4] followed by its nodes [2^levels] x int4 (zero padded)
        if (grd_ok_ && kern::obl2_feasible(static_cast<int>(MD), static_cast<int>(D), true)) {
            const size_t MX = kern::obl2_levels(static_cast<int>(MD)), DMAX = kern::obl2_padded_outputs(static_cast<int>(D)), DW = DMAX / 4;
            const size_t LS = size_t(1) << MX, VT = LS * DMAX, RECF = VT + LS * 4;   // floats (= dwords) per record
            constexpr size_t kPadTrees = 32;
            const size_t vpad = kPadTrees * RECF + (size_t(64) << 10) / sizeof(float);
            if (up_trees_ == 0) values_sw_host_.clear();
            values_sw_host_.resize(T * RECF + vpad);
            std::fill(values_sw_host_.begin() + static_cast<long>(up_trees_ * RECF), values_sw_host_.end(), 0.0f);
            for (size_t t = up_trees_; t < T; ++t) {
                float *rec = &values_sw_host_[t * RECF];
                const size_t l0 = static_cast<size_t>(model.tree_indices[t]), l1 = t + 1 < T ? static_cast<size_t>(model.tree_indices[t + 1]) : L;
                for (size_t leaf = 0; leaf < l1 - l0 && leaf < LS; ++leaf)
                    for (size_t j = 0; j < D; ++j) rec[((j / DW) * LS + leaf) * DW + (j % DW)] = model.values[(l0 + leaf) * D + j];
                const size_t n0 = static_cast<size_t>(grd_off_host_[t]), n1 = static_cast<size_t>(grd_off_host_[t + 1]);
                if (n1 - n0 <= LS) std::memcpy(rec + VT, &grd_nodes_host_[n0 * 4], (n1 - n0) * 16);
            }
            append_padded(m_values_sw_, values_sw_host_.data(), up_trees_, T, RECF, vpad);
        }
    }
    up_trees_ = T; up_leaves_ = L; up_splits_ = S;
    // small, mutable state: always refreshed
    append(m_bias_, model.bias.data(), 4, 0, D);
    std::vector<int32_t> os, oe;
    std::vector<float> olr;
    for (const auto &o : model.opts) { os.push_back(o.start_idx); oe.push_back(o.stop_idx); olr.push_back(o.init_lr); }  // ConstScheduler::get_lr
    append(m_opt_start_, os.data(), 4, 0, os.size());
    append(m_opt_stop_, oe.data(), 4, 0, oe.size());
    append(m_opt_lr_, olr.data(), 4, 0, olr.size());
    flush_segments();
    mirror_version_ = model.version;
}


// The code book of the packed-code predict path (kern::predict_pc, predict_reg.hip).  Every numeric condition `x > t` of the
// ensemble becomes `field < (0xffff - rank(t)) << 16`, where the field of a row is 0xffff - #{distinct thresholds of the feature
// below x}; every categorical condition `cell == category` becomes "bit `slot` of the row's inverted one-hot words is clear".
// Conditions that can never pass (threshold +inf or NaN, the padding in front of a shallow tree) get T = 0.  Rebuilt from the
// host model whenever it has changed (a new threshold shifts the ranks of its feature).
bool Engine::ensure_pc_book(int n_num, int n_cat) {
    if (pc_version_ == model.version && pc_f_ == n_num && pc_fc_ == n_cat) return pc_ok_;
    pc_version_ = model.version; pc_f_ = n_num; pc_fc_ = n_cat; pc_ok_ = false;
    hipStream_t s = stream_;
    const gbrl_hip_metadata &md = model.meta;
    const size_t T = md.n_trees, MD = md.max_depth;
    const size_t MX = kern::obl2_levels(static_cast<int>(MD));
    if (MX == 0 || T == 0) return false;
    auto never = [](float v) { return v != v || v == std::numeric_limits<float>::infinity(); };
    std::vector<std::vector<float>> thr(n_num);
    std::vector<std::vector<int32_t>> ids(n_cat);
    for (size_t t = 0; t < T; ++t) {
        const size_t depth = static_cast<size_t>(model.depths[t]);
        for (size_t d = 0; d < depth; ++d) {
            const size_t c = t * MD + d;
            const int f = model.feature_indices[c];
            if (model.is_numerics[c]) {
                if (f < 0 || f >= n_num) return false;
                const float v = model.feature_values[c];
                if (!never(v)) thr[f].push_back(v == 0.0f ? 0.0f : v);   // -0 and +0 are one threshold
            } else {
                if (f < 0 || f >= n_cat || cat_ids_host_[c] <= 0) return false;
                ids[f].push_back(cat_ids_host_[c]);
            }
        }
    }
    std::vector<int32_t> thr_off(n_num + 1, 0);
    std::vector<float> thr_all;
    size_t max_m = 1;
    for (int f = 0; f < n_num; ++f) {
        std::sort(thr[f].begin(), thr[f].end());
        thr[f].erase(std::unique(thr[f].begin(), thr[f].end()), thr[f].end());
        if (thr[f].size() > 65534) return false;
        max_m = std::max(max_m, thr[f].size());
        thr_all.insert(thr_all.end(), thr[f].begin(), thr[f].end());
        thr_off[f + 1] = static_cast<int32_t>(thr_all.size());
    }
    thr_all.push_back(0.0f);   // never empty
    int iters = 1;
    while ((size_t(1) << iters) <= max_m) ++iters;   // 2^iters > max_m: the descent can reach every count 0..max_m
    std::vector<int32_t> cat_slot(cat_dict_.size() + 1, -1);
    std::vector<int32_t> col_first_slot(n_cat + 1, 0);
    int n_slots = 0;
    for (int c = 0; c < n_cat; ++c) {
        std::sort(ids[c].begin(), ids[c].end());
        ids[c].erase(std::unique(ids[c].begin(), ids[c].end()), ids[c].end());
        col_first_slot[c] = n_slots;
        for (int32_t id : ids[c]) {
            if (id >= static_cast<int32_t>(cat_slot.size())) return false;
            cat_slot[id] = n_slots++;
        }
    }
    col_first_slot[n_cat] = n_slots;
    const int wn = (n_num + 1) / 2, cw = (n_slots + 31) / 32, nw = wn + cw, row_words = (nw + 3) & ~3;
    if (row_words > kern::predict_pc_bank_words() || row_words == 0) return false;
    std::vector<int32_t> word_cols(2 * std::max(cw, 1), 0);
    for (int q = 0; q < cw; ++q) {
        int first = n_cat, last = 0;
        for (int c = 0; c < n_cat; ++c)
            if (col_first_slot[c + 1] > 32 * q && col_first_slot[c] < 32 * (q + 1) && col_first_slot[c + 1] > col_first_slot[c]) { first = std::min(first, c); last = std::max(last, c + 1); }
        word_cols[2 * q] = first; word_cols[2 * q + 1] = std::max(first, last);
    }
    constexpr size_t kPadTrees = 32;
    std::vector<int32_t> rec((T + kPadTrees) * MX * 3, 0);
    for (size_t t = 0; t < T; ++t) {
        const size_t depth = static_cast<size_t>(model.depths[t]);
        if (depth > MX) return false;
        int32_t *r = &rec[(t * MX + (MX - depth)) * 3];
        for (size_t d = 0; d < depth; ++d, r += 3) {
            const size_t c = t * MD + d;
            const int f = model.feature_indices[c];
            if (model.is_numerics[c]) {
                const float v0 = model.feature_values[c];
                if (never(v0)) continue;   // (0, 0, 0): never true
                const float v = v0 == 0.0f ? 0.0f : v0;
                const size_t k = static_cast<size_t>(std::lower_bound(thr[f].begin(), thr[f].end(), v) - thr[f].begin());
                r[0] = f >> 1;
                r[1] = (f & 1) ? 0 : 16;
                r[2] = static_cast<int32_t>(static_cast<uint32_t>(0xffffu - k) << 16);
            } else {
                const int slot = cat_slot[cat_ids_host_[c]];
                r[0] = wn + (slot >> 5);
                r[1] = 31 - (slot & 31);
                r[2] = static_cast<int32_t>(0x80000000u);
            }
        }
    }
    auto up = [&](DevBuf &b, const void *src, size_t bytes) {
        hip_check(hipMemcpyAsync(b.ensure(bytes), src, bytes, hipMemcpyHostToDevice, s), "H2D code book");
    };
    up(m_pc_cond_, rec.data(), rec.size() * 4);
    up(m_pc_thr_, thr_all.data(), thr_all.size() * 4);
    up(m_pc_thr_off_, thr_off.data(), thr_off.size() * 4);
    up(m_pc_cat_slot_, cat_slot.data(), cat_slot.size() * 4);
    up(m_pc_word_cols_, word_cols.data(), word_cols.size() * 4);
    hip_check(hipStreamSynchronize(s), "sync code book");   // the host vectors go out of scope
    pc_wn_ = wn; pc_nw_ = nw; pc_row_words_ = row_words; pc_iters_ = iters;
    pc_ok_ = true;
    return true;
}

// Dictionary ids of a batch of categorical cells (0 = a category no condition of the model mentions).  Needs sync_model_to_device().
int32_t *Engine::encode_categorical_batch(const char *cat, bool cat_dev, int n, int n_cat) {
    hipStream_t s = stream_;
    int32_t *dcat = nullptr;
    // dictionary encoding on the device: the cells are hashed (strcmp semantics: bytes before the first NUL), looked up in
    // the per-feature, hash-sorted dictionary of the categories the model's conditions mention, and confirmed word by word
    const char *dcells = cat;
    if (!cat_dev) {
        char *tmp = static_cast<char *>(d_pcells_.ensure(static_cast<size_t>(n) * n_cat * kCat));
        hip_check(hipMemcpyAsync(tmp, cat, static_cast<size_t>(n) * n_cat * kCat, hipMemcpyHostToDevice, s), "H2D cat cells");
        dcells = tmp;
    }
    if (dict_version_ != cat_dict_.size() || dict_fc_ != n_cat) {
        struct E { uint64_t h; int id; uint64_t w[16]; };
        std::vector<std::vector<E>> per(n_cat);
        for (size_t z = 0; z < cat_dict_.size(); ++z) {
            const int f = cat_dict_[z].first;
            if (f < 0 || f >= n_cat) continue;
            E e{};
            uint64_t raw[16];
            std::memcpy(raw, cat_dict_[z].second.data(), kCat);
            e.h = cat_cell_hash(raw, e.w);
            e.id = static_cast<int>(z) + 1;
            per[f].push_back(e);
        }
        std::vector<int32_t> off(n_cat + 1, 0), ids;
        std::vector<uint64_t> hs, ws;
        for (int f = 0; f < n_cat; ++f) {
            std::sort(per[f].begin(), per[f].end(), [](const E &a, const E &b) { return a.h < b.h || (a.h == b.h && a.id < b.id); });
            for (const E &e : per[f]) { hs.push_back(e.h); ids.push_back(e.id); ws.insert(ws.end(), e.w, e.w + 16); }
            off[f + 1] = static_cast<int32_t>(hs.size());
        }
        hs.push_back(0); ids.push_back(0); ws.resize(ws.size() + 16, 0);   // never empty
        hip_check(hipMemcpyAsync(d_dict_off_.ensure(off.size() * 4), off.data(), off.size() * 4, hipMemcpyHostToDevice, s), "H2D dict");
        hip_check(hipMemcpyAsync(d_dict_hash_.ensure(hs.size() * 8), hs.data(), hs.size() * 8, hipMemcpyHostToDevice, s), "H2D dict");
        hip_check(hipMemcpyAsync(d_dict_id_.ensure(ids.size() * 4), ids.data(), ids.size() * 4, hipMemcpyHostToDevice, s), "H2D dict");
        hip_check(hipMemcpyAsync(d_dict_words_.ensure(ws.size() * 8), ws.data(), ws.size() * 8, hipMemcpyHostToDevice, s), "H2D dict");
        hip_check(hipStreamSynchronize(s), "sync");   // the host vectors go out of scope
        dict_version_ = cat_dict_.size();
        dict_fc_ = n_cat;
    }
    dcat = static_cast<int32_t *>(d_pcat_.ensure(sizeof(int32_t) * static_cast<size_t>(n) * n_cat));
    kern::encode_categories(dcells, n, n_cat, d_dict_off_.as<int32_t>(), d_dict_hash_.as<uint64_t>(), d_dict_id_.as<int32_t>(),
                            d_dict_words_.as<uint64_t>(), dcat, s);
    return dcat;
}

uint64_t Engine::cat_dict_token() {
    if (dict_token_size_ != cat_dict_.size()) {
        uint64_t h = 0xcbf29ce484222325ull;      // FNV-1a over (feature, 128-byte category) in dictionary order
        auto mix = [&](const void *p, size_t nbytes) { const unsigned char *b = static_cast<const unsigned char *>(p); for (size_t i = 0; i < nbytes; ++i) { h ^= b[i]; h *= 0x100000001b3ull; } };
        for (const auto &e : cat_dict_) {
            const int32_t f = e.first;
            char cell[kCat] = {0};
            std::memcpy(cell, e.second.data(), std::min<size_t>(kCat, e.second.size()));
            mix(&f, sizeof(f));
            mix(cell, kCat);
        }
        const uint64_t n_entries = cat_dict_.size();
        mix(&n_entries, sizeof(n_entries));
        dict_token_ = h;
        dict_token_size_ = cat_dict_.size();
    }
    return dict_token_;
}

void Engine::encode_categorical(const char *cat, bool cat_dev, int n, int n_cat, int32_t *ids_out, bool out_dev, uint64_t *token) {
    gbrl_hip_metadata &md = model.meta;
    if (n <= 0 || n_cat <= 0 || cat == nullptr || ids_out == nullptr) throw InvalidArgument("encode_categorical: no cells");
    if (md.iteration != 0 && n_cat != md.n_cat_features) throw InvalidArgument("Incompatible dataset");
    ensure_device();
    sync_model_to_device();
    hipStream_t s = stream_;
    const int32_t *dcat = encode_categorical_batch(cat, cat_dev, n, n_cat);
    hip_check(hipMemcpyAsync(ids_out, dcat, sizeof(int32_t) * static_cast<size_t>(n) * n_cat, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s), "ids out");
    hip_check(hipStreamSynchronize(s), "sync");
    if (token) *token = cat_dict_token();
}

void Engine::predict(const float *obs, bool obs_dev, const char *cat, bool cat_dev, int n, int n_num, int n_cat, int start_tree,
                     int stop_tree, float *out, bool out_dev) {
    predict_core(obs, obs_dev, cat, cat_dev, nullptr, false, nullptr, n, n_num, n_cat, start_tree, stop_tree, out, out_dev);
}

void Engine::predict_encoded(const float *obs, bool obs_dev, const int32_t *cat_ids, bool ids_dev, uint64_t token, int n, int n_num, int n_cat,
                             int start_tree, int stop_tree, float *out, bool out_dev) {
    if (n_cat > 0 && cat_ids == nullptr) throw InvalidArgument("Cannot call predict without observations!");
    predict_core(obs, obs_dev, nullptr, false, cat_ids, ids_dev, &token, n, n_num, n_cat, start_tree, stop_tree, out, out_dev);
}

void Engine::predict_core(const float *obs, bool obs_dev, const char *cat, bool cat_dev, const int32_t *cat_ids, bool ids_dev, const uint64_t *token,
                          int n, int n_num, int n_cat, int start_tree, int stop_tree, float *out, bool out_dev) {
    gbrl_hip_metadata &md = model.meta;
    // GBRL::predict, gbrl.cpp:378-390
    if (md.iteration == 0) { md.n_num_features = n_num; md.n_cat_features = n_cat; }
    if (n_num + n_cat != md.input_dim) throw InvalidArgument("Incompatible dataset");
    if (n_num != md.n_num_features || n_cat != md.n_cat_features) throw InvalidArgument("Incompatible dataset");
    if (n <= 0 || out == nullptr) throw InvalidArgument("Cannot call predict without observations!");
    if (n_num > 0 && obs == nullptr) throw InvalidArgument("Cannot call predict without observations!");
    if (n_cat > 0 && cat == nullptr && cat_ids == nullptr) throw InvalidArgument("Cannot call predict without observations!");
    if (md.output_dim > 128) throw Unsupported("predict: output_dim > 128");
    if (start_tree < 0 || stop_tree < 0) throw InvalidArgument("invalid tree range");   // the reference would index out of bounds
    ensure_device();
    ev_used_ = 0;
    ev_names_.clear();
    hipStream_t s = stream_;
    const int D = md.output_dim;
    // predict_cpu, predictor.cpp:127-141
    int stop = stop_tree;
    if (md.n_trees == 0 || stop > md.n_trees || model.opts.empty()) { start_tree = 0; stop = 0; }
    else if (stop == 0) stop = md.n_trees;
    // an empty or inverted range walks no tree: predict_cpu's loops run from start to stop (predictor.cpp:139-163), the result is the bias
    if (start_tree >= stop) { start_tree = 0; stop = 0; }
    sync_model_to_device();
    phase_begin();
    const float *dobs = obs;
    if (n_num > 0 && !obs_dev) {
        dobs = static_cast<float *>(d_pobs_.ensure(sizeof(float) * static_cast<size_t>(n) * n_num));
        hip_check(hipMemcpyAsync(const_cast<float *>(dobs), obs, sizeof(float) * static_cast<size_t>(n) * n_num, hipMemcpyHostToDevice, s), "H2D obs");
    }
    int32_t *dcat = nullptr;
    if (n_cat > 0 && cat_ids != nullptr) {
        // pre-encoded ids (encode_categorical): valid only for the dictionary they were made from
        if (token == nullptr || *token != cat_dict_token())
            throw InvalidArgument("predict: the categorical ids were encoded for another category dictionary (the model has grown or is a different one): encode the batch again");
        if (ids_dev) {
            dcat = const_cast<int32_t *>(cat_ids);
        } else {
            dcat = static_cast<int32_t *>(d_pcat_in_.ensure(sizeof(int32_t) * static_cast<size_t>(n) * n_cat));
            hip_check(hipMemcpyAsync(dcat, cat_ids, sizeof(int32_t) * static_cast<size_t>(n) * n_cat, hipMemcpyHostToDevice, s), "H2D cat ids");
        }
    } else if (n_cat > 0) {
        dcat = encode_categorical_batch(cat, cat_dev, n, n_cat);
    }
    float *dout = out;
    if (!out_dev) dout = static_cast<float *>(d_pout_.ensure(sizeof(float) * static_cast<size_t>(n) * D));
    phase_end("inputs");
    phase_begin(/*key=*/true);
    kern::PredictModel pm{};
    pm.tree_indices = m_tree_indices_.as<int32_t>();
    pm.depths = m_depths_.as<int32_t>();
    pm.feature_indices = m_feature_indices_.as<int32_t>();
    pm.cat_ids = m_cat_ids_.as<int32_t>();
    pm.feature_values = m_feature_values_.as<float>();
    pm.values = m_values_.as<float>();
    pm.bias = m_bias_.as<float>();
    pm.is_numerics = m_is_numerics_.as<uint8_t>();
    pm.inequality_directions = m_ineq_.as<uint8_t>();
    pm.n_trees = md.n_trees; pm.n_leaves = md.n_leaves; pm.max_depth = md.max_depth; pm.D = D;
    pm.oblivious = model.oblivious() ? 1 : 0;
    pm.n_opts = static_cast<int>(model.opts.size());
    pm.opt_start = m_opt_start_.as<int32_t>();
    pm.opt_stop = m_opt_stop_.as<int32_t>();
    pm.opt_lr = m_opt_lr_.as<float>();
    pm.cond_pack = m_cond_pack_.as<int32_t>();
    pm.grd_nodes = m_grd_nodes_.as<int32_t>();
    pm.grd_node_off = m_grd_off_.as<int32_t>();
    pm.grd_ok = (!model.oblivious() && grd_ok_) ? 1 : 0;
    pm.grd_max_nodes = grd_max_nodes_;
    pm.grd_max_leaves = grd_max_leaves_;
    pm.obl_ok = model.oblivious() ? 1 : 0;
    pm.obl2_maxd = 0; pm.values_sw = nullptr; pm.cond_ra = nullptr;
    pm.cat_dict_size = static_cast<int>(cat_dict_.size());
    if (model.oblivious() && kern::obl2_feasible(md.max_depth, D, false) && md.n_trees > 0) {
        pm.obl2_maxd = kern::obl2_levels(md.max_depth);
        pm.values_sw = m_values_sw_.as<float>();
        pm.cond_ra = m_cond_ra_.as<int32_t>();
    } else if (!model.oblivious() && grd_ok_ && kern::obl2_feasible(md.max_depth, D, true) && md.n_trees > 0) {
        pm.obl2_maxd = kern::obl2_levels(md.max_depth);   // greedy mode of the same kernel: records = values + nodes
        pm.values_sw = m_values_sw_.as<float>();
    }
    // Packed-code path: large batches of rows the fp32 register-tile kernel does not take (categorical columns, more than 128 or an
    // odd number of features, unaligned rows).  The code book follows the model; the rows are packed inside kern::predict.
    pm.pc_cond = nullptr; pm.pc_rows = nullptr;
    {
        const char *no_pc = hooks::raw(hooks::PREDICT_NO_PC), *no_reg = hooks::raw(hooks::PREDICT_NO_REG), *mr = hooks::raw(hooks::PREDICT_REG_MIN_ROWS);
        const int min_rows = mr ? std::atoi(mr) : 32768;
        const bool fp32_takes_it = n_cat == 0 && n_num <= 128 && (n_num & 3) == 0 && (reinterpret_cast<uintptr_t>(dobs) & 15) == 0 && !(no_reg && no_reg[0] == '1');
        if (!(no_pc && no_pc[0] == '1') && !in_fit_ && pm.values_sw != nullptr && model.oblivious() && n >= min_rows && !fp32_takes_it &&
            kern::predict_pc_shape_ok(pm.obl2_maxd, D) && ensure_pc_book(n_num, n_cat)) {
            pm.pc_cond = m_pc_cond_.as<int32_t>();
            pm.pc_thr = m_pc_thr_.as<float>();
            pm.pc_thr_off = m_pc_thr_off_.as<int32_t>();
            pm.pc_cat_slot = m_pc_cat_slot_.as<int32_t>();
            pm.pc_word_cols = m_pc_word_cols_.as<int32_t>();
            pm.pc_wn = pc_wn_; pm.pc_nw = pc_nw_; pm.pc_row_words = pc_row_words_; pm.pc_iters = pc_iters_;
            pm.pc_rows = static_cast<uint32_t *>(d_pc_rows_.ensure(static_cast<size_t>(n) * pc_row_words_ * sizeof(uint32_t)));
        }
    }
    if (const char *e = hooks::raw(hooks::PREDICT_OBL1)) {      // test / measurement hook: the first-generation oblivious kernel
        if (e[0] == '1') pm.obl2_maxd = 0;
    }
    if (const char *e = hooks::raw(hooks::PREDICT_GENERIC)) {   // test hook: the general kernels only
        if (e[0] == '1') { pm.grd_ok = 0; pm.obl_ok = 0; }
    }
    pm.coef_ok = D <= 64 ? 1 : 0;
    pm.coef_cover = 0;
    for (int j = 0; j < 64; ++j) pm.coef[j] = 0.0f;
    for (const auto &o : model.opts) {   // one learning rate per output unless two optimisers share an output
        for (int j = o.start_idx; j < o.stop_idx && pm.coef_ok; ++j) {
            if (j < 0 || j >= D || ((pm.coef_cover >> j) & 1ull)) { pm.coef_ok = 0; break; }
            pm.coef_cover |= 1ull << j;
            pm.coef[j] = o.init_lr;
        }
    }
    // small batches: scratch for up to 64 partial sums per output (tree ranges spread over blocks, kern::predict)
    pm.partial = nullptr; pm.partial_floats = 0; pm.tree_chunk = 0;
    // (not inside fit(): its gradients follow the reference's per-row tree-order chain at every batch size)
    pm.par_th = md.par_th;
    // (nor for a model whose file cleared parallel_predict: the reference then runs the chain for every batch, predictor.cpp:144)
    const char *nosplit = hooks::raw(hooks::PREDICT_NOSPLIT);   // measurement hook: no partial sums over tree ranges
    if (!in_fit_ && model.parallel_predict && n <= 64 * 256 && stop - start_tree >= 128 && !(nosplit && nosplit[0] == '1')) {
        pm.partial_floats = std::min<size_t>(static_cast<size_t>(64) * n * D, size_t(16) << 20);
        pm.partial = static_cast<float *>(d_pred_partial_.ensure(pm.partial_floats * sizeof(float)));
    }
    // Up to 8192 rows against a large tree range: leaf search spread over the chip + one multiply-add chain per (row, output)
    // (kern::predict_chain: the bits of the one-chain-per-row kernels, so fit() uses it too).  Measured against the tiled kernels'
    // 80 ns per tree: 1024 rows x 20 000 trees 1.59 -> 0.27 ms, 4096 rows 1.58 -> 0.64 ms; beyond ~8192 rows the leaf search
    // alone costs what the tiled kernel does.  GBRL_HIP_PREDICT_CHAIN=0 / 1: never / whenever the shape is covered (tests).
    pm.slots = nullptr; pm.slot_ints = 0;
    {
        const int trees = stop - start_tree;
        bool want = (n <= kChainMaxRows && trees >= kChainMinTrees) || (n <= 1024 && trees >= 128);
        if (const char *e = hooks::raw(hooks::PREDICT_CHAIN)) want = e[0] == '1' ? (n <= 64 * 256 && trees >= 1) : false;
        if (want) {
            const size_t ints = kern::predict_chain_slot_ints(n, trees);
            if (ints <= (size_t(1) << 27)) {
                pm.slot_ints = ints;
                pm.slots = static_cast<int32_t *>(d_pred_slots_.ensure(ints * sizeof(int32_t)));
            }
        }
    }
    kern::predict(pm, dobs, n_num, dcat, n_cat, n, start_tree, stop, dout, s);
    hip_check(hipGetLastError(), "predict launch");
    phase_end("predict", /*key=*/true);
    if (!out_dev) hip_check(hipMemcpyAsync(out, dout, sizeof(float) * static_cast<size_t>(n) * D, hipMemcpyDeviceToHost, s), "D2H preds");
    hip_check(hipStreamSynchronize(s), "sync");
    phases_resolve();
}

}  // namespace gbrl
