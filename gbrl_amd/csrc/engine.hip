// engine.hip -- Engine infrastructure: grow-only device / pinned buffers, device selection, the exchange points of the
// row-sharded mode (native RCCL or caller hooks) and the per-phase timing events.  The step / fit orchestration is in
// engine_step.hip, model mirroring and predict in engine_predict.hip (see engine.h).
#include "engine.h"
#include "hooks.h"

#include <numeric>
#include <random>
#include <functional>
#include "rccl_dyn.h"
#include "cat_hash.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <unordered_map>

namespace gbrl {

using kern::Chunk;
using kern::FeatureSlot;
using kern::NodeSplit;

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw HipError(std::string(what) + ": " + hipGetErrorString(e));
}

DevBuf::~DevBuf() { release(); }
void DevBuf::release() {
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr;
    cap_ = 0;
}
void *DevBuf::ensure(size_t bytes) {
    if (bytes <= cap_ && ptr_) return ptr_;
    release();
    const size_t want = std::max<size_t>(256, bytes + bytes / 8);
    hip_check(hipMalloc(&ptr_, want), "hipMalloc");
    cap_ = want;
    return ptr_;
}
PinnedBuf::~PinnedBuf() {
    if (ptr_) (void)hipHostFree(ptr_);
}
void *PinnedBuf::ensure(size_t bytes) {
    if (bytes <= cap_ && ptr_) return ptr_;
    if (ptr_) (void)hipHostFree(ptr_);
    ptr_ = nullptr;
    const size_t want = std::max<size_t>(4096, bytes + bytes / 4);
    // polled by the host while kernels store into it (level results, leaf sums): fine-grained coherent + device-mapped, stated explicitly
    hip_check(hipHostMalloc(&ptr_, want, hipHostMallocCoherent | hipHostMallocMapped), "hipHostMalloc");
    cap_ = want;
    return ptr_;
}
void *DevBuf::ensure_keep(size_t bytes, size_t keep_bytes, hipStream_t s) {
    if (bytes <= cap_ && ptr_) return ptr_;
    void *np = nullptr;
    const size_t want = std::max<size_t>(4096, bytes * 2);
    hip_check(hipMalloc(&np, want), "hipMalloc");
    if (ptr_ && keep_bytes) {
        hip_check(hipMemcpyAsync(np, ptr_, keep_bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync(grow)");
        hip_check(hipStreamSynchronize(s), "hipStreamSynchronize(grow)");
    }
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = np;
    cap_ = want;
    return ptr_;
}

Engine::Engine(const gbrl_hip_config &cfg) : model(cfg), device_ordinal_(cfg.device_ordinal) {}
Engine::Engine(Model &&loaded, int device_ordinal) : model(std::move(loaded)), device_ordinal_(device_ordinal) {}
Engine::Engine(const Engine &o) : model(o.model), device_ordinal_(o.device_ordinal_) {}

Engine::~Engine() {
    if (device_ready_) {
        (void)hipSetDevice(device_ordinal_);
        for (auto &e : ev_pool_) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (rccl_comm_ && rccl_api().ok) (void)rccl_api().CommDestroy(rccl_comm_);
        if (ev_level_) (void)hipEventDestroy(ev_level_);
        if (ev_model_stage_) (void)hipEventDestroy(ev_model_stage_);
        if (own_stream_) (void)hipStreamDestroy(own_stream_);
    }
}

void Engine::ensure_device() {
    if (device_ready_) {
        hip_check(hipSetDevice(device_ordinal_), "hipSetDevice");
        return;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw NoDeviceError("no HIP device available: this library has no CPU path (the reference's CPU path lives in oracle/)");
    if (device_ordinal_ < 0) {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) cur = 0;
        device_ordinal_ = cur;
    }
    if (device_ordinal_ >= count) throw InvalidArgument("device ordinal out of range");
    hip_check(hipSetDevice(device_ordinal_), "hipSetDevice");
    hip_check(hipEventCreateWithFlags(&ev_level_, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&ev_model_stage_, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipStreamCreate(&own_stream_), "hipStreamCreate");  // blocking stream: ordered with the null stream torch uses by default
    if (stream_ == nullptr) stream_ = own_stream_;
    device_ready_ = true;
}

int Engine::device_ordinal() {
    ensure_device();
    return device_ordinal_;
}

void Engine::set_stream(hipStream_t s) {
    ensure_device();
    if (stream_) hip_check(hipStreamSynchronize(stream_), "hipStreamSynchronize(set_stream)");   // nothing of ours is left on the old stream
    stream_ = s ? s : own_stream_;
}

void Engine::set_collective(const gbrl_hip_collective *hooks) {
    // world_size 1 normally means "no exchange"; GBRL_HIP_FORCE_COLLECTIVE=1 keeps the hooks installed anyway so that the
    // sharded code path (hook calls, stream hand-over, no sibling subtraction, counting quantiles) can be tested on ONE GPU
    const char *force = gbrl::hooks::raw(gbrl::hooks::FORCE_COLLECTIVE);
    if (rccl_comm_ && rccl_api().ok) { (void)rccl_api().CommDestroy(rccl_comm_); rccl_comm_ = nullptr; }
    if (hooks == nullptr || (hooks->world_size <= 1 && !(force && force[0] == '1'))) {
        has_coll_ = false;
        return;
    }
    if (!hooks->allreduce_sum_i64 || !hooks->allreduce_sum_f64 || !hooks->allreduce_max_f32 || !hooks->allreduce_min_f32)
        throw InvalidArgument("collective hooks incomplete");
    coll_ = *hooks;
    has_coll_ = true;
}

void Engine::set_rccl(const void *id128, int world_size, int rank, bool keep_world1) {
    const RcclApi &api = rccl_api();
    if (!api.ok) throw Unsupported("RCCL is not available in this process");
    if (world_size < 1 || rank < 0 || rank >= world_size || id128 == nullptr) throw InvalidArgument("invalid RCCL communicator arguments");
    ensure_device();
    if (rccl_comm_) { (void)api.CommDestroy(rccl_comm_); rccl_comm_ = nullptr; }
    RcclApi::UniqueId id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    RcclApi::Comm comm = nullptr;
    const int rc = api.CommInitRank(&comm, world_size, id, rank);
    if (rc != 0 || comm == nullptr) throw HipError(std::string("ncclCommInitRank failed: ") + (api.GetErrorString ? api.GetErrorString(rc) : "?"));
    rccl_comm_ = comm;
    coll_ = gbrl_hip_collective{};
    coll_.world_size = world_size;
    coll_.rank = rank;
    // world size 1 normally means "no exchange"; the caller's flag (GBRL_HIP_RCCL_KEEP_WORLD1, bench.py's `collective` leg) or the diagnostic
    // hook keeps the row-sharded code path anyway
    has_coll_ = world_size > 1 || keep_world1 || hooks::on(hooks::FORCE_COLLECTIVE);
}

// One exchange point: in-place all-reduce of a device buffer.  With an RCCL communicator the call is enqueued on the engine's
// stream and the host does not wait; with caller-provided hooks the stream is drained first (the hook's contract).
void Engine::reduce_scatter_i64(int64_t *send, int64_t *recv, size_t count) {
    const int P = coll_.world_size, r = coll_.rank;
    ++exch_calls_;
    if (rccl_comm_ && rccl_api().ReduceScatter) {
        exch_bytes_ += count * sizeof(int64_t) * static_cast<size_t>(P) / 2;   // a reduce-scatter moves half of what the all-reduce of its send buffer moves
        const int rc = rccl_api().ReduceScatter(send, recv, count, RcclApi::kInt64, RcclApi::kSum, rccl_comm_, stream_);
        if (rc != 0) throw HipError(std::string("ncclReduceScatter failed: ") + (rccl_api().GetErrorString ? rccl_api().GetErrorString(rc) : "?"));
        return;
    }
    --exch_calls_;
    exch_bytes_ -= count * sizeof(int64_t) * static_cast<size_t>(P) / 2;   // hooks: counted as the reduce-scatter it stands for
    exchange(Red::SumI64, send, count * static_cast<size_t>(P));
    hip_check(hipMemcpyAsync(recv, send + static_cast<size_t>(r) * count, count * sizeof(int64_t), hipMemcpyDeviceToDevice, stream_), "D2D own slice");
}

void Engine::exchange(Red op, void *dev_buf, size_t count) {
    ++exch_calls_;
    exch_bytes_ += count * (op == Red::MaxF32 || op == Red::MinF32 ? 4 : 8);      // all-reduce payload
    if (rccl_comm_) {
        const RcclApi &api = rccl_api();
        int dt = RcclApi::kInt64, ro = RcclApi::kSum;
        switch (op) {
            case Red::SumI64: dt = RcclApi::kInt64; ro = RcclApi::kSum; break;
            case Red::SumF64: dt = RcclApi::kFloat64; ro = RcclApi::kSum; break;
            case Red::MaxF32: dt = RcclApi::kFloat32; ro = RcclApi::kMax; break;
            case Red::MinF32: dt = RcclApi::kFloat32; ro = RcclApi::kMin; break;
        }
        const int rc = api.AllReduce(dev_buf, dev_buf, count, dt, ro, rccl_comm_, stream_);
        if (rc != 0) throw HipError(std::string("ncclAllReduce failed: ") + (api.GetErrorString ? api.GetErrorString(rc) : "?"));
        return;
    }
    hip_check(hipStreamSynchronize(stream_), "sync before exchange");
    int rc = -1;
    switch (op) {
        case Red::SumI64: rc = coll_.allreduce_sum_i64(coll_.ctx, static_cast<int64_t *>(dev_buf), count); break;
        case Red::SumF64: rc = coll_.allreduce_sum_f64(coll_.ctx, static_cast<double *>(dev_buf), count); break;
        case Red::MaxF32: rc = coll_.allreduce_max_f32(coll_.ctx, static_cast<float *>(dev_buf), count); break;
        case Red::MinF32: rc = coll_.allreduce_min_f32(coll_.ctx, static_cast<float *>(dev_buf), count); break;
    }
    if (rc != 0) throw HipError("allreduce failed");
}
int Engine::radix_exchange_trampoline(void *self, int64_t *dev_buf, size_t count) {
    try {
        static_cast<Engine *>(self)->exchange(Red::SumI64, dev_buf, count);
        return 0;
    } catch (...) {
        return 2;
    }
}

// Phase timing: HIP events recorded on the model's stream WITHOUT synchronising; resolved once at the end of the call
// (after the call's final stream synchronisation), so enabling it does not perturb the timed region.
void Engine::phase_begin(bool key) {
    if (profiling_ < (key ? 1 : 2)) return;
    if (ev_used_ == ev_pool_.size()) {
        hipEvent_t a, b;
        hip_check(hipEventCreate(&a), "hipEventCreate");
        hip_check(hipEventCreate(&b), "hipEventCreate");
        ev_pool_.push_back({a, b});
    }
    hip_check(hipEventRecord(ev_pool_[ev_used_].first, stream_), "hipEventRecord");
}
void Engine::phase_end(const char *name, bool key) {
    // a failed launch is sticky and would otherwise surface at the next checked call: attribute it to the phase that made it
    { const hipError_t e = hipGetLastError(); if (e != hipSuccess) throw HipError(std::string("kernel launch failed in phase '") + name + "': " + hipGetErrorString(e)); }
    if (profiling_ < (key ? 1 : 2)) return;
    hip_check(hipEventRecord(ev_pool_[ev_used_].second, stream_), "hipEventRecord");
    ev_names_.push_back(name);
    ++ev_used_;
}
std::pair<hipEvent_t, hipEvent_t> Engine::kernel_events(const char *name, bool key) {
    if (profiling_ < (key ? 1 : 2)) return {nullptr, nullptr};
    // Level 1 SAMPLES the key kernel: one launch in seven carries the event pair, so that over consecutive steps every tree level is
    // measured in turn (7 is coprime to the usual 6 levels).  A pair on every launch cost 5.5 us each -- 1.8 % of a 2^20 x 128 step.
    if (profiling_ == 1 && key && (key_calls_++ % 7u) != 0u) return {nullptr, nullptr};
    if (ev_used_ == ev_pool_.size()) {
        hipEvent_t a, b;
        hip_check(hipEventCreate(&a), "hipEventCreate");
        hip_check(hipEventCreate(&b), "hipEventCreate");
        ev_pool_.push_back({a, b});
    }
    ev_names_.push_back(name);
    return ev_pool_[ev_used_++];
}
void Engine::phases_resolve() {
    phases_.clear();
    if (!profiling_) return;
    for (size_t i = 0; i < ev_used_; ++i) {
        float ms = 0.f;
        hip_check(hipEventSynchronize(ev_pool_[i].second), "hipEventSynchronize");
        hip_check(hipEventElapsedTime(&ms, ev_pool_[i].first, ev_pool_[i].second), "hipEventElapsedTime");
        bool found = false;
        for (auto &p : phases_)
            if (p.first == ev_names_[i]) { p.second += ms; found = true; break; }
        if (!found) phases_.emplace_back(ev_names_[i], ms);
    }
    if (profiling_ == 1) {   // how many launches the sampled total stands for (not a time)
        float n = 0.f;
        for (size_t i = 0; i < ev_used_; ++i) n += std::strcmp(ev_names_[i], "hist_build") == 0 ? 1.f : 0.f;
        phases_.emplace_back("hist_build_sampled_launches", n);
    }
    ev_used_ = 0;
    ev_names_.clear();
    if (profiling_ >= 2 && prep_launches_ > 0) phases_.emplace_back("prep_launches", static_cast<float>(prep_launches_));   // not a time: 1 = the fused preparation kernel ran
    if (profiling_ >= 2) {   // not times: counters since the engine was created (near-tie replay, neartie.hip)
        phases_.emplace_back("near_replays", static_cast<float>(near_replays_));
        phases_.emplace_back("near_bailouts", static_cast<float>(near_bailouts_));
        phases_.emplace_back("small_grow_fallbacks", static_cast<float>(small_grow_fallbacks_));
        phases_.emplace_back("cat_clash_redos", static_cast<float>(cat_clash_redos_));
        phases_.emplace_back("near_in_kernel", static_cast<float>(near_in_kernel_));
    }
    if (has_coll_) {   // not times: what this call handed to the transport (all-reduce payload; a reduce-scatter counts half its send buffer)
        phases_.emplace_back("exchange_payload_mb", static_cast<float>(exch_bytes_ / 1e6));
        phases_.emplace_back("exchange_calls", static_cast<float>(exch_calls_));
    }
}
}  // namespace gbrl
