// c_api.cpp -- the extern "C" surface declared in include/gbrl_hip.h.  Exceptions never cross the boundary: they are
// turned into status codes + a thread-local message (the reference throws std::runtime_error at the same places).
#include <mutex>
#include <unordered_map>
#include <vector>
#include <utility>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <new>
#include <string>

#include "../../include/gbrl_hip.h"
#include "engine.h"
#include "hooks.h"
#include "explain.h"
#include "rccl_dyn.h"

struct gbrl_hip_model {
    gbrl::Engine engine;
    std::vector<const char *> phase_names;
    explicit gbrl_hip_model(const gbrl_hip_config &c) : engine(c) {}
    gbrl_hip_model(gbrl::Model &&m, int dev) : engine(std::move(m), dev) {}
    explicit gbrl_hip_model(const gbrl_hip_model &o) : engine(o.engine) {}
};

namespace {
thread_local std::string g_err;

template <typename Fn>
int guarded(Fn &&fn) {
    gbrl::hooks::begin_call();   // environment hooks are read at most once per API call (hooks.h)
    try {
        fn();
        return GBRL_HIP_OK;
    } catch (const gbrl::NoDeviceError &e) { g_err = e.what(); return GBRL_HIP_E_NO_DEVICE;
    } catch (const gbrl::HipError &e) { g_err = e.what(); return GBRL_HIP_E_HIP;
    } catch (const gbrl::Unsupported &e) { g_err = e.what(); return GBRL_HIP_E_UNSUPPORTED;
    } catch (const gbrl::InvalidArgument &e) { g_err = e.what(); return GBRL_HIP_E_INVALID;
    } catch (const std::bad_alloc &) { g_err = "out of host memory"; return GBRL_HIP_E_INVALID;
    } catch (const std::exception &e) { g_err = e.what(); return GBRL_HIP_E_INVALID; }
}

int copy_in(void *dst, const void *src, size_t bytes, int on_device) {
    if (!on_device) { std::memcpy(dst, src, bytes); return 0; }
    return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
}  // namespace

// Output buffers of predict() on the device are recycled: hipMalloc + hipFree (which synchronises the device) cost about
// a millisecond per call, more than the prediction kernel itself for small ensembles.  A freed buffer is kept (up to a few
// buffers / 1 GiB) and handed out again for a request of the same size class after a device synchronisation, so that no
// consumer kernel of its previous life can still be reading it.
namespace {
struct DevPool {
    struct Buf { void *ptr; size_t cap; int device; };
    std::mutex mu;
    std::unordered_map<void *, Buf> live;               // every buffer handed out
    std::vector<Buf> idle;                              // freed buffers kept for reuse ON THEIR OWN DEVICE
    size_t idle_bytes = 0;
    static constexpr size_t kMaxIdleBytes = size_t(1) << 30;
    static constexpr size_t kMaxIdle = 8;
    ~DevPool() { /* process exit: the runtime reclaims device memory; calling hipFree here can race with its teardown */ }
};
DevPool &pool() { static DevPool *p = new DevPool(); return *p; }

// RAII: make `device` current for the calling thread, restore the previous one afterwards
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (device >= 0 && device != prev) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace


extern "C" {

int gbrl_hip_abi_version(void) { return GBRL_HIP_ABI_VERSION; }

int gbrl_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *gbrl_hip_last_error(void) { return g_err.c_str(); }

void *gbrl_hip_device_alloc_on(int device, size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) { g_err = "no HIP device"; return nullptr; }
    DeviceScope scope(device);
    DevPool &P = pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (size_t i = 0; i < P.idle.size(); ++i) {
            if (P.idle[i].device == device && P.idle[i].cap >= bytes && P.idle[i].cap <= bytes + bytes / 4 + 4096) {
                const DevPool::Buf b = P.idle[i];
                P.idle_bytes -= b.cap;
                P.idle.erase(P.idle.begin() + static_cast<long>(i));
                P.live[b.ptr] = b;
                (void)hipDeviceSynchronize();   // of `device` (current in this scope): no consumer of the buffer's previous life is still reading it
                return b.ptr;
            }
        }
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        // memory pressure: drop this device's idle buffers and retry once
        std::vector<DevPool::Buf> drop;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            for (size_t i = 0; i < P.idle.size();) {
                if (P.idle[i].device == device) { drop.push_back(P.idle[i]); P.idle_bytes -= P.idle[i].cap; P.idle.erase(P.idle.begin() + static_cast<long>(i)); }
                else ++i;
            }
        }
        for (auto &d : drop) (void)hipFree(d.ptr);
        if (hipMalloc(&p, bytes) != hipSuccess) { g_err = "hipMalloc failed"; return nullptr; }
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[p] = DevPool::Buf{p, bytes, device};
    return p;
}
void *gbrl_hip_device_alloc(size_t bytes) { return gbrl_hip_device_alloc_on(-1, bytes); }
void gbrl_hip_device_free(void *ptr) {
    if (!ptr) return;
    DevPool &P = pool();
    int device = -1;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(ptr);
        if (it != P.live.end()) {
            const DevPool::Buf b = it->second;
            device = b.device;
            P.live.erase(it);
            if (P.idle.size() < DevPool::kMaxIdle && P.idle_bytes + b.cap <= DevPool::kMaxIdleBytes) {
                P.idle.push_back(b);
                P.idle_bytes += b.cap;
                return;
            }
        }
    }
    DeviceScope scope(device);
    (void)hipFree(ptr);
}

int gbrl_hip_device_ordinal(gbrl_hip_model *m) {
    int dev = -1;
    (void)guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        dev = m->engine.device_ordinal();
    });
    return dev;
}

int gbrl_hip_set_stream(gbrl_hip_model *m, void *hip_stream) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        m->engine.set_stream(static_cast<hipStream_t>(hip_stream));
    });
}

gbrl_hip_model *gbrl_hip_create(const gbrl_hip_config *cfg) {
    gbrl_hip_model *m = nullptr;
    guarded([&] {
        if (!cfg) throw gbrl::InvalidArgument("null config");
        if (cfg->input_dim <= 0 || cfg->output_dim <= 0 || cfg->max_depth < 0 || cfg->max_depth > 30 || cfg->n_bins <= 0)
            throw gbrl::InvalidArgument("invalid model dimensions");
        if (cfg->use_control_variates)
            throw gbrl::Unsupported("control variates are CPU-only in the reference (gbrl.cpp:204-207) and not part of this build");
        if (cfg->split_score_func != GBRL_HIP_SCORE_L2 && cfg->split_score_func != GBRL_HIP_SCORE_COSINE)
            throw gbrl::InvalidArgument("Invalid score function! Options are: Cosine/L2");
        if (cfg->generator_type != GBRL_HIP_GEN_UNIFORM && cfg->generator_type != GBRL_HIP_GEN_QUANTILE)
            throw gbrl::InvalidArgument("Invalid generator function! Options are: Uniform/Quantile");
        if (cfg->grow_policy != GBRL_HIP_GROW_GREEDY && cfg->grow_policy != GBRL_HIP_GROW_OBLIVIOUS)
            throw gbrl::InvalidArgument("Invalid generator function! Options are: Greedy/Oblivious");
        m = new gbrl_hip_model(*cfg);
    });
    return m;
}

gbrl_hip_model *gbrl_hip_clone(const gbrl_hip_model *other) {
    gbrl_hip_model *m = nullptr;
    guarded([&] {
        if (!other) throw gbrl::InvalidArgument("null model");
        m = new gbrl_hip_model(*other);
    });
    return m;
}

gbrl_hip_model *gbrl_hip_load(const char *filename) {
    gbrl_hip_model *m = nullptr;
    int rc = guarded([&] {
        if (!filename) throw gbrl::InvalidArgument("null filename");
        m = new gbrl_hip_model(gbrl::Model::load(filename), -1);
    });
    if (rc != 0 && g_err.find("file") != std::string::npos) { /* message already set */ }
    return m;
}

int gbrl_hip_save(gbrl_hip_model *m, const char *filename) {
    int rc = guarded([&] {
        if (!m || !filename) throw gbrl::InvalidArgument("null argument");
        m->engine.model.save(filename);
    });
    return rc == GBRL_HIP_OK ? rc : GBRL_HIP_E_IO;
}

void gbrl_hip_destroy(gbrl_hip_model *m) { delete m; }

int gbrl_hip_set_bias(gbrl_hip_model *m, const float *bias, int n, int on_device) {
    return guarded([&] {
        if (!m || !bias) throw gbrl::InvalidArgument("null argument");
        if (n != m->engine.model.meta.output_dim) throw gbrl::InvalidArgument("Incompatible dimensions");
        if (copy_in(m->engine.model.bias.data(), bias, sizeof(float) * n, on_device)) throw gbrl::HipError("hipMemcpy failed");
        ++m->engine.model.version;
    });
}

int gbrl_hip_set_feature_weights(gbrl_hip_model *m, const float *w, int n, int on_device) {
    return guarded([&] {
        if (!m || !w) throw gbrl::InvalidArgument("null argument");
        if (n != m->engine.model.meta.input_dim) throw gbrl::InvalidArgument("Incompatible dimensions");
        if (copy_in(m->engine.model.feature_weights.data(), w, sizeof(float) * n, on_device)) throw gbrl::HipError("hipMemcpy failed");
        ++m->engine.model.version;
    });
}

int gbrl_hip_set_feature_mapping(gbrl_hip_model *m, const int32_t *feature_mapping, const uint8_t *mapping_numerics, int n) {
    return guarded([&] {
        if (!m || !feature_mapping || !mapping_numerics) throw gbrl::InvalidArgument("null argument");
        if (n != m->engine.model.meta.input_dim) throw gbrl::InvalidArgument("Incompatible dimensions");
        m->engine.model.set_feature_mapping(feature_mapping, mapping_numerics);
    });
}

int gbrl_hip_set_optimizer(gbrl_hip_model *m, const gbrl_hip_optimizer *opt) {
    return guarded([&] {
        if (!m || !opt) throw gbrl::InvalidArgument("null argument");
        m->engine.model.add_optimizer(*opt);
    });
}

int gbrl_hip_get_metadata(const gbrl_hip_model *m, gbrl_hip_metadata *out) {
    if (!m || !out) return GBRL_HIP_E_INVALID;
    *out = m->engine.model.meta;
    return GBRL_HIP_OK;
}
int gbrl_hip_get_bias(const gbrl_hip_model *m, float *out) {
    if (!m || !out) return GBRL_HIP_E_INVALID;
    std::memcpy(out, m->engine.model.bias.data(), sizeof(float) * m->engine.model.bias.size());
    return GBRL_HIP_OK;
}
int gbrl_hip_get_feature_weights(const gbrl_hip_model *m, float *out) {
    if (!m || !out) return GBRL_HIP_E_INVALID;
    std::memcpy(out, m->engine.model.feature_weights.data(), sizeof(float) * m->engine.model.feature_weights.size());
    return GBRL_HIP_OK;
}
int gbrl_hip_get_feature_mapping(const gbrl_hip_model *m, int32_t *fm, uint8_t *mn) {
    if (!m) return GBRL_HIP_E_INVALID;
    const auto &md = m->engine.model;
    if (fm) std::memcpy(fm, md.feature_mapping.data(), sizeof(int32_t) * md.feature_mapping.size());
    if (mn) std::memcpy(mn, md.mapping_numerics.data(), md.mapping_numerics.size());
    return GBRL_HIP_OK;
}
int gbrl_hip_num_optimizers(const gbrl_hip_model *m) { return m ? static_cast<int>(m->engine.model.opts.size()) : 0; }
int gbrl_hip_get_optimizer(const gbrl_hip_model *m, int idx, gbrl_hip_optimizer *out) {
    if (!m || !out || idx < 0 || idx >= static_cast<int>(m->engine.model.opts.size())) return GBRL_HIP_E_INVALID;
    *out = m->engine.model.opts[idx];
    return GBRL_HIP_OK;
}
const char *gbrl_hip_learner_name(const gbrl_hip_model *m) { return m ? m->engine.model.learner_name.c_str() : ""; }

int gbrl_hip_get_ensemble(const gbrl_hip_model *m, int32_t *tree_indices, int32_t *depths, float *values,
                          int32_t *feature_indices, float *feature_values, float *edge_weights, uint8_t *is_numerics,
                          uint8_t *inequality_directions, char *categorical_values, int32_t *rev_num, int32_t *rev_cat) {
    if (!m) return GBRL_HIP_E_INVALID;
    const gbrl::Model &md = m->engine.model;
    const size_t T = md.meta.n_trees, L = md.meta.n_leaves, S = md.split_rows(), MD = md.meta.max_depth, D = md.meta.output_dim;
    auto cp = [](void *dst, const void *src, size_t bytes) { if (dst && bytes) std::memcpy(dst, src, bytes); };
    cp(tree_indices, md.tree_indices.data(), T * 4);
    cp(depths, md.depths.data(), S * 4);
    cp(values, md.values.data(), L * D * 4);
    cp(feature_indices, md.feature_indices.data(), S * MD * 4);
    cp(feature_values, md.feature_values.data(), S * MD * 4);
    cp(edge_weights, md.edge_weights.data(), L * MD * 4);
    cp(is_numerics, md.is_numerics.data(), S * MD);
    cp(inequality_directions, md.inequality_directions.data(), L * MD);
    cp(categorical_values, md.categorical_values.data(), S * MD * gbrl::kCat);
    cp(rev_num, md.reverse_num.data(), md.reverse_num.size() * 4);
    cp(rev_cat, md.reverse_cat.data(), md.reverse_cat.size() * 4);
    return GBRL_HIP_OK;
}

int gbrl_hip_step(gbrl_hip_model *m, const float *obs, int obs_on_device, const char *cat_obs, int cat_on_device,
                  const float *grads, int grads_on_device, int n_samples, int n_num_features, int n_cat_features) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        m->engine.step(obs, obs_on_device != 0, cat_obs, cat_on_device != 0, grads, grads_on_device != 0, n_samples,
                       n_num_features, n_cat_features);
    });
}

int gbrl_hip_fit(gbrl_hip_model *m, const float *obs, int obs_on_device, const char *cat_obs, int cat_on_device,
                 const float *targets, int targets_on_device, int n_samples, int n_num_features, int n_cat_features,
                 int iterations, int shuffle, float *loss_out) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        const float loss = m->engine.fit(obs, obs_on_device != 0, cat_obs, cat_on_device != 0, targets, targets_on_device != 0, n_samples,
                                         n_num_features, n_cat_features, iterations, shuffle != 0);
        if (loss_out) *loss_out = loss;
    });
}

int gbrl_hip_predict(gbrl_hip_model *m, const float *obs, int obs_on_device, const char *cat_obs, int cat_on_device,
                     int n_samples, int n_num_features, int n_cat_features, int start_tree, int stop_tree, float *out,
                     int out_on_device) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        m->engine.predict(obs, obs_on_device != 0, cat_obs, cat_on_device != 0, n_samples, n_num_features, n_cat_features,
                          start_tree, stop_tree, out, out_on_device != 0);
    });
}

int gbrl_hip_encode_categorical(gbrl_hip_model *m, const char *cat_obs, int cat_on_device, int n_samples, int n_cat_features, int32_t *ids_out,
                                int ids_on_device, uint64_t *dictionary_token) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        m->engine.encode_categorical(cat_obs, cat_on_device != 0, n_samples, n_cat_features, ids_out, ids_on_device != 0, dictionary_token);
    });
}

int gbrl_hip_predict_encoded(gbrl_hip_model *m, const float *obs, int obs_on_device, const int32_t *cat_ids, int ids_on_device,
                             uint64_t dictionary_token, int n_samples, int n_num_features, int n_cat_features, int start_tree, int stop_tree,
                             float *out, int out_on_device) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        m->engine.predict_encoded(obs, obs_on_device != 0, cat_ids, ids_on_device != 0, dictionary_token, n_samples, n_num_features, n_cat_features,
                                  start_tree, stop_tree, out, out_on_device != 0);
    });
}

int gbrl_hip_set_collective(gbrl_hip_model *m, const gbrl_hip_collective *hooks) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        m->engine.set_collective(hooks);
    });
}

int gbrl_hip_rccl_available(void) { return gbrl::rccl_api().ok ? 1 : 0; }
int gbrl_hip_rccl_unique_id(void *id128) {
    return guarded([&] {
        if (!id128) throw gbrl::InvalidArgument("null id buffer");
        const gbrl::RcclApi &api = gbrl::rccl_api();
        if (!api.ok) throw gbrl::Unsupported("RCCL is not available in this process");
        gbrl::RcclApi::UniqueId id;
        if (api.GetUniqueId(&id) != 0) throw gbrl::HipError("ncclGetUniqueId failed");
        std::memcpy(id128, id.internal, sizeof(id.internal));
    });
}
int gbrl_hip_set_rccl(gbrl_hip_model *m, const void *id128, int world_size, int rank) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        m->engine.set_rccl(id128, world_size, rank);
    });
}

int gbrl_hip_set_rccl_flags(gbrl_hip_model *m, const void *id128, int world_size, int rank, unsigned flags) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null model");
        if (flags & ~static_cast<unsigned>(GBRL_HIP_RCCL_KEEP_WORLD1)) throw gbrl::InvalidArgument("unknown RCCL flags");
        m->engine.set_rccl(id128, world_size, rank, (flags & GBRL_HIP_RCCL_KEEP_WORLD1) != 0);
    });
}

// ---- inspection (explain.cpp), served from the host copy of the ensemble ----
int gbrl_hip_tree_shap(const gbrl_hip_model *m, int tree_idx, const float *obs, const char *cat_obs, int n_samples,
                       const float *norm_values, const float *base_poly, const float *offset, float *out) {
    return guarded([&] {
        if (!m || !out || n_samples < 0) throw gbrl::InvalidArgument("null argument");
        const gbrl::Model &md = m->engine.model;
        gbrl::check_shap_arguments(md, tree_idx, obs, cat_obs, norm_values, base_poly, offset);   // validate BEFORE anything is written
        std::memset(out, 0, sizeof(float) * static_cast<size_t>(n_samples) * (md.meta.n_num_features + md.meta.n_cat_features) * md.meta.output_dim);
        // the device evaluates the same recursion with the same roundings (shap.hip); the host copy serves a machine without a GPU
        if (const_cast<gbrl_hip_model *>(m)->engine.shap_on_device(tree_idx, obs, cat_obs, n_samples, norm_values, base_poly, offset, out)) return;
        gbrl::tree_shap(md, tree_idx, obs, cat_obs, n_samples, norm_values, base_poly, offset, out);
    });
}
int gbrl_hip_ensemble_shap(const gbrl_hip_model *m, const float *obs, const char *cat_obs, int n_samples,
                           const float *norm_values, const float *base_poly, const float *offset, float *out) {
    return guarded([&] {
        if (!m || !out || n_samples < 0) throw gbrl::InvalidArgument("null argument");
        const gbrl::Model &md = m->engine.model;
        if (md.meta.n_trees == 0) return;   // nothing to explain; the caller's buffer is sized from its inputs (binding zero-fills)
        gbrl::check_shap_arguments(md, 0, obs, cat_obs, norm_values, base_poly, offset);
        std::memset(out, 0, sizeof(float) * static_cast<size_t>(n_samples) * (md.meta.n_num_features + md.meta.n_cat_features) * md.meta.output_dim);
        if (const_cast<gbrl_hip_model *>(m)->engine.shap_on_device(-1, obs, cat_obs, n_samples, norm_values, base_poly, offset, out)) return;
        gbrl::ensemble_shap(md, obs, cat_obs, n_samples, norm_values, base_poly, offset, out);
    });
}
int gbrl_hip_export(const gbrl_hip_model *m, const char *filename, const char *modelname, const char *export_format,
                    const char *export_type, const char *prefix) {
    bool io_error = false;
    const int rc = guarded([&] {
        if (!m || !filename) throw gbrl::InvalidArgument("null argument");
        // the reference opens (and truncates) the file before it validates anything (gbrl.cpp:1107-1118)
        FILE *f = std::fopen(filename, "wb");
        if (!f) { io_error = true; throw std::runtime_error("File opening error"); }
        std::string text;
        try {
            gbrl::export_header(m->engine.model, modelname ? modelname : "", export_format ? export_format : "float",
                                export_type ? export_type : "full", prefix ? prefix : "", text);
        } catch (...) { std::fclose(f); throw; }
        const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
        if (std::fclose(f) != 0 || !ok) { io_error = true; throw std::runtime_error("Writing to file error"); }
    });
    return (rc != GBRL_HIP_OK && io_error) ? GBRL_HIP_E_IO : rc;
}
int gbrl_hip_print_tree(const gbrl_hip_model *m, int tree_idx) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null argument");
        const std::string t = gbrl::tree_text(m->engine.model, tree_idx);
        std::fwrite(t.data(), 1, t.size(), stdout);
        std::fflush(stdout);
    });
}
int gbrl_hip_print_ensemble_metadata(const gbrl_hip_model *m, const char *device_name) {
    return guarded([&] {
        if (!m) throw gbrl::InvalidArgument("null argument");
        const std::string t = gbrl::metadata_text(m->engine.model, device_name ? device_name : "cuda");
        std::fwrite(t.data(), 1, t.size(), stdout);
        std::fflush(stdout);
    });
}
int gbrl_hip_plot_tree(const gbrl_hip_model *, int, const char *) {
    return guarded([&] { throw gbrl::Unsupported("GBRL compiled without Graphviz! Cannot plot model"); });
}
size_t gbrl_hip_alloc_data_size(const gbrl_hip_model *m) { return m ? gbrl::reference_alloc_bytes(m->engine.model, true) : 0; }

int gbrl_hip_last_phase_times(const gbrl_hip_model *m, const char **names, float *ms, int cap) {
    if (!m) return 0;
    const auto &ph = m->engine.phase_times();
    const int n = static_cast<int>(ph.size());
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = ph[i].first.c_str();
        if (ms) ms[i] = ph[i].second;
    }
    return n;
}

int gbrl_hip_replay_scores(const float *grads, const uint8_t *in_node, const uint8_t *goes_right, int n_rows, int output_dim,
                           const float *meanden, int cosine, int min_data_in_leaf, float *out_scores) {
    if (!grads || !in_node || !goes_right || !out_scores || n_rows < 1 || output_dim < 1) return GBRL_HIP_E_INVALID;
    return guarded([&] {
        if (!gbrl::kern::near_tie_selftest(grads, in_node, goes_right, n_rows, output_dim, meanden, cosine != 0, min_data_in_leaf, out_scores))
            throw gbrl::Unsupported("near-tie replay: shape not supported (n_rows <= 65536, output_dim <= 1024) or no HIP device");
    });
}

int gbrl_hip_seq_sums(const float *x, const uint32_t *lens, const float *starts, int n_chains, float *out, uint32_t *n_slow_blocks) {
    if (!lens || !out || n_chains < 0) return GBRL_HIP_E_INVALID;
    return guarded([&] {
        if (!gbrl::kern::seq_sums_selftest(x, lens, starts, n_chains, out, n_slow_blocks)) throw gbrl::HipError("sequential-sum kernels failed (no HIP device?)");
    });
}

int gbrl_hip_set_profiling(gbrl_hip_model *m, int enabled) {
    if (!m) return GBRL_HIP_E_INVALID;
    m->engine.set_profiling(enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled));
    return GBRL_HIP_OK;
}

}  // extern "C"
