// engine_step_detail.h -- what the three translation units of step() share (round 6: engine_step.hip was one 2 500-line file):
//   engine_candidates.hip  split candidates: categorical cells on the device, sharded ranking, numeric thresholds (A3-A5)
//   engine_grow.hip        Engine::grow_tree: the one-launch growth of RL-sized steps and the level loop (A6-A10)
//   engine_step.hip        Engine::step / Engine::fit, the host scan of categorical cells, the tree joining the ensemble (A1, A2, A10-A11)
#pragma once
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

namespace detail {

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

// GBRL_HIP_DEVICE_LEVELS=1 (opt-in device-planned level loop)
inline bool device_levels_requested() { return hooks::on(hooks::DEVICE_LEVELS); }
// Host side of the copy-free hand-overs (engine_candidates.hip)
void spin_until_published(volatile uint32_t *flag, uint32_t seq, hipStream_t s, const char *what);
// split_candidate_generator.cpp:216-249 (engine_candidates.hip)
std::vector<int64_t> quantile_target_ranks(long long n_global, int B);
// A5 on the host / A10-A11 (engine_step.hip)
void categorical_candidates(const char *hcat, const float *hgrads, int N, int Fc, int D, int B, std::vector<CatCandidate> &cat_cands,
                            std::vector<uint16_t> &h_catcodes, std::vector<int> &cat_classes);
void append_tree(Model &model, const std::vector<HNode> &nodes, const std::vector<int> &frontier, const std::vector<int64_t> &acc,
                 double leaf_scale, const std::vector<CatCandidate> &cat_cands);

}  // namespace detail

using detail::Stager;
using detail::device_levels_requested;
using detail::spin_until_published;
using detail::quantile_target_ranks;
using detail::categorical_candidates;
using detail::append_tree;

}  // namespace gbrl
