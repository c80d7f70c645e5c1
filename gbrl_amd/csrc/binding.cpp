// binding.cpp -- Python extension module `gbrl_cpp` exposing class `GBRL`, written against include/gbrl_hip.h ONLY.
//
// It mirrors the reference's Python-visible operator interface for the hot path (gbrl/src/cpp/binding.cpp:421-1131):
// same constructor keywords and defaults, same method names, same argument conventions (NumPy arrays or the 4-tuples
// (data_ptr, shape, dtype, device) that gbrl/common/utils.py:43-60 builds from torch tensors), same shape inference for
// 1-D inputs, same RuntimeError conditions, same return conventions (NumPy on "cpu", DLPack capsule on "cuda").
// Differences that are deliberate:
//   * every computation runs on the GPU; `device` only selects how predict() delivers its result.
//   * the DLPack capsule carries kDLROCM (10): torch-ROCm rejects the reference's hard-coded kDLCUDA.
//   * export / SHAP / print are served from the host copy of the ensemble (csrc/explain.cpp); plot_tree raises the
//     reference's own no-Graphviz error (this image has no Graphviz).
#include <pybind11/numpy.h>
#include <algorithm>
#include <pybind11/pybind11.h>

#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/gbrl_hip.h"

namespace py = pybind11;

namespace {

// ---- DLPack ABI (public spec, "dltensor" capsule protocol) -- the minimal declarations needed to hand a buffer over
enum { kDLCPU_ = 1, kDLROCM_ = 10 };
struct DLDevice_ { int32_t device_type; int32_t device_id; };
struct DLDataType_ { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DLTensor_ {
    void *data; DLDevice_ device; int32_t ndim; DLDataType_ dtype; int64_t *shape; int64_t *strides; uint64_t byte_offset;
};
struct DLManagedTensor_ {
    DLTensor_ dl_tensor; void *manager_ctx; void (*deleter)(DLManagedTensor_ *);
};

void dl_deleter(DLManagedTensor_ *self) {
    if (self->dl_tensor.device.device_type == kDLROCM_) gbrl_hip_device_free(self->dl_tensor.data);
    else delete[] static_cast<float *>(self->dl_tensor.data);
    delete[] self->dl_tensor.shape;
    delete self;
}

void capsule_destructor(PyObject *cap) {
    // an unconsumed capsule still owns the tensor (consumers rename it to "used_dltensor")
    if (PyCapsule_IsValid(cap, "dltensor")) {
        auto *mt = static_cast<DLManagedTensor_ *>(PyCapsule_GetPointer(cap, "dltensor"));
        if (mt && mt->deleter) mt->deleter(mt);
    }
}

py::object make_dlpack(void *data, const std::vector<int64_t> &shape, bool on_device, int device_id, bool int32 = false) {
    auto *mt = new DLManagedTensor_;
    mt->dl_tensor.data = data;
    mt->dl_tensor.device = {on_device ? kDLROCM_ : kDLCPU_, on_device ? device_id : 0};
    mt->dl_tensor.ndim = static_cast<int32_t>(shape.size());
    mt->dl_tensor.dtype = {static_cast<uint8_t>(int32 ? 0 /*kDLInt*/ : 2 /*kDLFloat*/), 32, 1};
    mt->dl_tensor.shape = new int64_t[shape.size()];
    std::copy(shape.begin(), shape.end(), mt->dl_tensor.shape);
    mt->dl_tensor.strides = nullptr;
    mt->dl_tensor.byte_offset = 0;
    mt->manager_ctx = nullptr;
    mt->deleter = dl_deleter;
    return py::reinterpret_steal<py::object>(PyCapsule_New(mt, "dltensor", capsule_destructor));
}

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error(msg); }
void check(int status) {
    if (status != GBRL_HIP_OK) fail(gbrl_hip_last_error());
}

struct Input {
    const void *ptr = nullptr;
    std::vector<size_t> shape;
    bool on_device = false;
    py::object keep;  // keeps a converted NumPy array alive for the duration of the call
};

// handle_input_info (binding.cpp:156-190): None | NumPy array | (data_ptr, shape, dtype, device)
// `categorical`: 0 = float32 values, 1 = 128-byte cells, 2 = int32 dictionary ids (encode_categorical / predict_encoded, an extension)
Input read_input(py::object &obj, const std::string &name, bool none_allowed, const std::string &fn, int categorical) {
    Input in;
    if (obj.is_none()) {
        if (!none_allowed) fail("Cannot call " + fn + " without " + name + "!");
        return in;
    }
    if (py::isinstance<py::array>(obj)) {
        py::array arr = py::array::ensure(obj, py::array::c_style | py::array::forcecast);
        if (!arr) fail("Could not convert object to a contiguous NumPy array");
        py::buffer_info info = arr.request();
        const std::string want = categorical == 1 ? "128s" : categorical == 2 ? py::format_descriptor<int32_t>::format() : py::format_descriptor<float>::format();
        if (info.format != want) {
            std::stringstream ss;
            ss << "Expected array of format '" << want << "', but got '" << info.format << "'";
            fail(ss.str());
        }
        in.ptr = info.ptr;
        in.shape.assign(info.shape.begin(), info.shape.end());
        in.keep = arr;
        return in;
    }
    if (py::isinstance<py::tuple>(obj)) {
        py::tuple t = obj.cast<py::tuple>();
        if (t.size() != 4) fail("Expected a tuple of size 4: (data_ptr, shape, dtype, device)");
        const uintptr_t raw = t[0].cast<uintptr_t>();
        in.ptr = (raw == 0 || raw == static_cast<uintptr_t>(-1)) ? nullptr : reinterpret_cast<const void *>(raw);
        for (py::handle d : t[1].cast<py::tuple>()) in.shape.push_back(d.cast<size_t>());
        const std::string dtype = t[2].cast<std::string>();
        if (categorical == 2) {
            if (dtype != "torch.int32") fail("Expected dtype torch.int32, but got " + dtype);
        } else if (categorical) {
            // extension over the reference (which takes categorical cells as NumPy S128 arrays only): a device-resident cell
            // matrix, [n, n_cat] cells of 128 bytes each (e.g. a torch.uint8 tensor [n, n_cat, 128]), announced as dtype "S128"
            if (dtype != "S128" && dtype != "|S128") fail("Unsupported data type: " + dtype);
        } else if (dtype != "torch.float32") {
            fail("Expected dtype torch.float32, but got " + dtype);
        }
        const std::string dev = t[3].cast<std::string>();
        if (dev == "cpu") in.on_device = false;
        else if (dev == "cuda" || dev == "gpu") in.on_device = true;
        else fail("Invalid device! Options are: cpu/cuda");
        return in;
    }
    fail("Unknown " + name + " type! Must be a NumPy array or tuple.");
}

int parse_enum(const std::string &s, std::initializer_list<std::pair<const char *, int>> opts, const char *err) {
    for (const auto &o : opts)
        if (s == o.first) return o.second;
    fail(err);
}
int parse_device(const std::string &s) {  // stringTodeviceType, types.cpp:58-62
    return parse_enum(s, {{"cpu", 0}, {"cuda", 1}, {"gpu", 1}}, "Invalid device! Options are: cpu/cuda");
}

class PyGBRL {
   public:
    gbrl_hip_model *h = nullptr;
    int device = 0;  // 0 = "cpu" delivery (NumPy), 1 = "cuda" delivery (DLPack / kDLROCM)

    PyGBRL(int input_dim, int output_dim, int policy_dim, int max_depth, int min_data_in_leaf, int n_bins, int par_th,
           float cv_beta, const std::string &split_score_func, const std::string &generator_type, bool use_cv,
           int batch_size, const std::string &grow_policy, int verbose, const std::string &dev, const std::string &name) {
        gbrl_hip_config c{};
        c.input_dim = input_dim; c.output_dim = output_dim; c.policy_dim = policy_dim; c.max_depth = max_depth;
        c.min_data_in_leaf = min_data_in_leaf; c.n_bins = n_bins; c.par_th = par_th; c.cv_beta = cv_beta;
        // string parsing as types.cpp:31-62
        c.split_score_func = parse_enum(split_score_func, {{"L2", 0}, {"l2", 0}, {"Cosine", 1}, {"cosine", 1}},
                                        "Invalid score function! Options are: Cosine/L2");
        c.generator_type = parse_enum(generator_type, {{"uniform", 0}, {"Uniform", 0}, {"quantile", 1}, {"Quantile", 1}},
                                      "Invalid generator function! Options are: Uniform/Quantile");
        c.grow_policy = parse_enum(grow_policy, {{"greedy", 0}, {"Greedy", 0}, {"oblivious", 1}, {"Oblivious", 1}},
                                   "Invalid generator function! Options are: Greedy/Oblivious");
        c.use_control_variates = use_cv ? 1 : 0;
        c.batch_size = batch_size; c.verbose = verbose; c.device_ordinal = -1; c.learner_name = name.c_str();
        device = parse_device(dev);
        h = gbrl_hip_create(&c);
        if (!h) fail(gbrl_hip_last_error());
    }
    explicit PyGBRL(const PyGBRL &o) : device(o.device) {
        h = gbrl_hip_clone(o.h);
        if (!h) fail(gbrl_hip_last_error());
    }
    explicit PyGBRL(gbrl_hip_model *loaded) : h(loaded), device(0) {}
    ~PyGBRL() { gbrl_hip_destroy(h); }

    gbrl_hip_metadata meta() const {
        gbrl_hip_metadata m{};
        gbrl_hip_get_metadata(h, &m);
        return m;
    }
};

py::object predict_impl(PyGBRL &self, py::object &obs, py::object &cat, py::object start_obj, py::object stop_obj,
                        bool return_torch, const uint64_t *ids_token = nullptr /* non-null: `cat` holds int32 dictionary ids (predict_encoded) */) {
    const gbrl_hip_metadata md = self.meta();
    const int start = start_obj.is_none() ? 0 : start_obj.cast<int>();
    const int stop = stop_obj.is_none() ? 0 : stop_obj.cast<int>();
    // bounds as binding.cpp:800-811
    if (start < 0 || (start >= md.n_trees && md.n_trees > 0)) {
        std::stringstream ss;
        ss << "start_tree_idx is out of bounds! Got " << start << ", but valid range is [0, " << md.n_trees - 1 << "]";
        fail(ss.str());
    }
    if (stop < 0 || stop > md.n_trees) {
        std::stringstream ss;
        ss << "stop_tree_idx is out of bounds! Got " << stop << ", but valid range is [0, " << md.n_trees << "]";
        fail(ss.str());
    }
    Input o = read_input(obs, "obs", true, "predict", false);
    Input c = read_input(cat, "cat_obs", true, "predict", ids_token ? 2 : 1);
    if (!o.ptr && !c.ptr) fail("Cannot call predict without observations!");
    int n = 0, n_num = 0, n_cat = 0;
    const int in_dim = md.input_dim;
    auto neq = [&](size_t a, size_t b) {
        if (a != b) {
            std::stringstream ss;
            ss << "Number of samples is not equal between obs and categorical obs " << a << " != " << b;
            fail(ss.str());
        }
    };
    // shape inference for 1-D inputs as binding.cpp:820-923
    if (o.ptr && c.ptr) {
        if (o.shape.size() == 1 && c.shape.size() == 1) {
            if (static_cast<int>(o.shape[0] + c.shape[0]) == in_dim) { n = 1; n_num = static_cast<int>(o.shape[0]); n_cat = static_cast<int>(c.shape[0]); }
            else { neq(o.shape[0], c.shape[0]); n = static_cast<int>(o.shape[0]); n_num = 1; n_cat = 1; }
        } else if (o.shape.size() == 1) { neq(o.shape[0], c.shape[0]); n = static_cast<int>(o.shape[0]); n_num = 1; n_cat = static_cast<int>(c.shape[1]);
        } else if (c.shape.size() == 1) { neq(o.shape[0], c.shape[0]); n = static_cast<int>(o.shape[0]); n_num = static_cast<int>(o.shape[1]); n_cat = 1;
        } else { neq(o.shape[0], c.shape[0]); n = static_cast<int>(o.shape[0]); n_num = static_cast<int>(o.shape[1]); n_cat = static_cast<int>(c.shape[1]); }
    } else if (o.ptr) {
        if (o.shape.size() == 1) {
            if (static_cast<int>(o.shape[0]) == in_dim) { n = 1; n_num = in_dim; } else { n = static_cast<int>(o.shape[0]); n_num = 1; }
        } else { n = static_cast<int>(o.shape[0]); n_num = static_cast<int>(o.shape[1]); }
    } else {
        if (c.shape.size() == 1) {
            if (static_cast<int>(c.shape[0]) == in_dim) { n = 1; n_cat = in_dim; } else { n = static_cast<int>(c.shape[0]); n_cat = 1; }
        } else { n = static_cast<int>(c.shape[0]); n_cat = static_cast<int>(c.shape[1]); }
    }
    if (n_num + n_cat != in_dim) {
        std::stringstream ss;
        ss << "Total number of features " << n_num + n_cat << " != input dim " << in_dim;
        fail(ss.str());
    }
    const int D = md.output_dim;
    std::vector<int64_t> shape;
    if (D == 1) shape = {n}; else shape = {n, D};  // binding.cpp:281-286
    const bool dev_out = self.device == 1;
    float *out = nullptr;
    int dev_id = 0;
    if (dev_out) {
        // the buffer lives on the MODEL's device (one rank per GPU: not necessarily device 0) and the capsule says so
        dev_id = gbrl_hip_device_ordinal(self.h);
        if (dev_id < 0) fail(gbrl_hip_last_error());
        out = static_cast<float *>(gbrl_hip_device_alloc_on(dev_id, sizeof(float) * static_cast<size_t>(n) * D));
        if (!out) fail(gbrl_hip_last_error());
    } else {
        out = new float[static_cast<size_t>(n) * D];
    }
    int rc;
    {
        py::gil_scoped_release release;  // binding.cpp:934
        if (ids_token)
            rc = gbrl_hip_predict_encoded(self.h, static_cast<const float *>(o.ptr), o.on_device, static_cast<const int32_t *>(c.ptr), c.on_device, *ids_token,
                                          n, n_num, n_cat, start, stop, out, dev_out);
        else
        rc = gbrl_hip_predict(self.h, static_cast<const float *>(o.ptr), o.on_device, static_cast<const char *>(c.ptr),
                              c.on_device, n, n_num, n_cat, start, stop, out, dev_out);
    }
    if (rc != GBRL_HIP_OK) {
        if (dev_out) gbrl_hip_device_free(out); else delete[] out;
        fail(gbrl_hip_last_error());
    }
    if (dev_out || return_torch) return make_dlpack(out, shape, dev_out, dev_id);
    py::capsule owner(out, [](void *p) { delete[] static_cast<float *>(p); });
    std::vector<py::ssize_t> shp(shape.begin(), shape.end());
    return py::array_t<float>(shp, out, owner);
}

// Extension: (ids, token) = encode_categorical(categorical_obs): int32 dictionary ids [n, n_cat] of a batch of cells -- a DLPack capsule on the
// model's device for a "cuda" model, a NumPy array otherwise -- for predict_encoded(obs, ids, token, ...).
py::tuple encode_categorical_impl(PyGBRL &self, py::object &cat) {
    Input c = read_input(cat, "cat_obs", false, "encode_categorical", 1);
    if (!c.ptr) fail("Cannot call encode_categorical without cat_obs!");
    const int n = static_cast<int>(c.shape[0]), n_cat = c.shape.size() > 1 ? static_cast<int>(c.shape[1]) : 1;
    const bool dev_out = self.device == 1;
    int32_t *ids = nullptr;
    int dev_id = 0;
    if (dev_out) {
        dev_id = gbrl_hip_device_ordinal(self.h);
        if (dev_id < 0) fail(gbrl_hip_last_error());
        ids = static_cast<int32_t *>(gbrl_hip_device_alloc_on(dev_id, sizeof(int32_t) * static_cast<size_t>(n) * n_cat));
        if (!ids) fail(gbrl_hip_last_error());
    } else {
        ids = new int32_t[static_cast<size_t>(n) * n_cat];
    }
    uint64_t token = 0;
    int rc;
    {
        py::gil_scoped_release release;
        rc = gbrl_hip_encode_categorical(self.h, static_cast<const char *>(c.ptr), c.on_device, n, n_cat, ids, dev_out, &token);
    }
    if (rc != GBRL_HIP_OK) {
        if (dev_out) gbrl_hip_device_free(ids); else delete[] ids;
        fail(gbrl_hip_last_error());
    }
    const std::vector<int64_t> shape = {n, n_cat};
    if (dev_out) return py::make_tuple(make_dlpack(ids, shape, true, dev_id, /*int32=*/true), token);
    py::capsule owner(ids, [](void *p) { delete[] static_cast<int32_t *>(p); });
    return py::make_tuple(py::array_t<int32_t>(std::vector<py::ssize_t>{n, n_cat}, ids, owner), token);
}

void step_impl(PyGBRL &self, py::object &obs, py::object &cat, py::object &grads) {
    const gbrl_hip_metadata md = self.meta();
    Input g = read_input(grads, "grads", false, "step", false);
    int n, gdim;
    // binding.cpp:462-478
    if (g.shape.size() == 1) {
        if (md.output_dim > 1) { n = 1; gdim = static_cast<int>(g.shape[0]); } else { n = static_cast<int>(g.shape[0]); gdim = 1; }
    } else { n = static_cast<int>(g.shape[0]); gdim = static_cast<int>(g.shape[1]); }
    if (gdim != md.output_dim) {
        std::stringstream ss;
        ss << "Gradient output dim " << gdim << " != correct output dim " << md.output_dim;
        fail(ss.str());
    }
    int n_num = 0, n_cat = 0;
    Input o = read_input(obs, "obs", true, "step", false);
    if (o.ptr) {  // binding.cpp:483-496
        int no;
        if (o.shape.size() == 1) { n_num = (n == 1) ? static_cast<int>(o.shape[0]) : 1; no = (n == 1) ? 1 : static_cast<int>(o.shape[0]); }
        else { no = static_cast<int>(o.shape[0]); n_num = static_cast<int>(o.shape[1]); }
        if (no != n) {
            std::stringstream ss;
            ss << "Number of observations " << no << " != number of gradient samples " << n;
            fail(ss.str());
        }
    }
    Input c = read_input(cat, "cat_obs", true, "step", true);
    if (c.ptr) {  // binding.cpp:502-515
        int nc;
        if (c.shape.size() == 1) { n_cat = (n == 1) ? static_cast<int>(c.shape[0]) : 1; nc = (n == 1) ? 1 : static_cast<int>(c.shape[0]); }
        else { nc = static_cast<int>(c.shape[0]); n_cat = static_cast<int>(c.shape[1]); }
        if (nc != n) {
            std::stringstream ss;
            ss << "Number of categorical observations " << nc << " != number of gradient samples " << n;
            fail(ss.str());
        }
    }
    if (n_cat + n_num != md.input_dim) {
        std::stringstream ss;
        ss << "Total number of features " << n_cat + n_num << " != correct input dim " << md.input_dim;
        fail(ss.str());
    }
    int rc;
    {
        py::gil_scoped_release release;  // binding.cpp:525
        rc = gbrl_hip_step(self.h, static_cast<const float *>(o.ptr), o.on_device, static_cast<const char *>(c.ptr), c.on_device,
                           static_cast<const float *>(g.ptr), g.on_device, n, n_num, n_cat);
    }
    check(rc);
}

float fit_impl(PyGBRL &self, py::object &obs, py::object &cat, py::object &targets, int iterations, bool shuffle,
               const std::string &loss_type) {
    if (loss_type != "MultiRMSE") fail("Invalid loss function! Options are: MultiRMSE");   // stringTolossType, types.cpp:52-56
    const gbrl_hip_metadata md = self.meta();
    Input t = read_input(targets, "targets", false, "fit", false);
    int n, tdim;   // binding.cpp:541-556
    if (t.shape.size() == 1) {
        if (md.output_dim > 1) { n = 1; tdim = static_cast<int>(t.shape[0]); } else { n = static_cast<int>(t.shape[0]); tdim = 1; }
    } else { n = static_cast<int>(t.shape[0]); tdim = static_cast<int>(t.shape[1]); }
    if (tdim != md.output_dim) {
        std::stringstream ss;
        ss << "Targets output dim " << tdim << " != correct output dim " << md.output_dim;
        fail(ss.str());
    }
    int n_num = 0, n_cat = 0;
    Input o = read_input(obs, "obs", true, "fit", false);
    if (o.ptr) {
        int no;
        if (o.shape.size() == 1) { n_num = (n == 1) ? static_cast<int>(o.shape[0]) : 1; no = (n == 1) ? 1 : static_cast<int>(o.shape[0]); }
        else { no = static_cast<int>(o.shape[0]); n_num = static_cast<int>(o.shape[1]); }
        if (no != n) {
            std::stringstream ss;
            ss << "Number of observations " << no << " != number of gradient samples " << n;
            fail(ss.str());
        }
    }
    Input c = read_input(cat, "cat_obs", true, "fit", true);
    if (c.ptr) {
        int nc;
        if (c.shape.size() == 1) { n_cat = (n == 1) ? static_cast<int>(c.shape[0]) : 1; nc = (n == 1) ? 1 : static_cast<int>(c.shape[0]); }
        else { nc = static_cast<int>(c.shape[0]); n_cat = static_cast<int>(c.shape[1]); }
        if (nc != n) {
            std::stringstream ss;
            ss << "Number of categorical observations " << nc << " != number of gradient samples " << n;
            fail(ss.str());
        }
    }
    if (n_cat + n_num != md.input_dim) {
        std::stringstream ss;
        ss << "Total number of features " << n_cat + n_num << " != correct input dim " << md.input_dim;
        fail(ss.str());
    }
    float loss = 0.0f;
    int rc;
    {
        py::gil_scoped_release release;
        rc = gbrl_hip_fit(self.h, static_cast<const float *>(o.ptr), o.on_device, static_cast<const char *>(c.ptr), c.on_device,
                          static_cast<const float *>(t.ptr), t.on_device, n, n_num, n_cat, iterations, shuffle ? 1 : 0, &loss);
    }
    check(rc);
    return loss;
}

// tree_shap / ensemble_shap (binding.cpp:985-1117): NumPy inputs only, result [n_samples][n_num + n_cat][output_dim]
struct HostArray {
    const void *ptr = nullptr;
    std::vector<py::ssize_t> shape;
    py::object keep;
};
template <typename ArrayT>
HostArray host_array(py::object &obj) {
    HostArray a;
    if (obj.is_none()) return a;
    ArrayT arr = py::cast<ArrayT>(obj);
    if (!arr.attr("flags").attr("c_contiguous").template cast<bool>()) fail("Arrays must be C-contiguous");
    py::buffer_info info = arr.request();
    a.ptr = info.ptr;
    a.shape.assign(info.shape.begin(), info.shape.end());
    a.keep = arr;
    return a;
}

py::array_t<float> shap_impl(PyGBRL &self, bool whole_ensemble, int tree_idx, py::object &obs, py::object &categorical_obs,
                             py::object &norm_values, py::object &base_poly, py::object &offset) {
    const HostArray o = host_array<py::array_t<float>>(obs);
    const HostArray c = host_array<py::array>(categorical_obs);
    const HostArray nv = host_array<py::array_t<float>>(norm_values), bp = host_array<py::array_t<float>>(base_poly),
                    of = host_array<py::array_t<float>>(offset);
    // 1-D inputs are ONE sample (binding.cpp:996-1003, 1014-1021)
    py::ssize_t n = 0, n_num = 0, n_cat = 0;
    if (o.ptr) { if (o.shape.size() == 1) { n_num = o.shape[0]; n = 1; } else { n_num = o.shape[1]; n = o.shape[0]; } }
    if (c.ptr) {
        if (c.shape.size() == 1) { n_cat = c.shape[0]; if (n == 0) n = 1; } else { n_cat = c.shape[1]; if (n == 0) n = c.shape[0]; }
    }
    const gbrl_hip_metadata md = self.meta();
    // the reference trusts the caller here and reads out of bounds on a mismatch; this build checks
    // (also when the ensemble is empty: predict() may already have latched the feature counts, and the library sizes what it writes from them)
    if ((md.n_trees > 0 || md.n_num_features + md.n_cat_features > 0) && (n_num != md.n_num_features || n_cat != md.n_cat_features))
        fail("Incompatible dimensions");
    const size_t depth = static_cast<size_t>(md.max_depth);
    auto count = [](const HostArray &a) { size_t k = a.ptr ? 1 : 0; for (py::ssize_t d : a.shape) k *= static_cast<size_t>(d); return k; };
    if (count(nv) < (depth + 1) * depth || count(bp) < depth || count(of) < depth * depth)
        fail("norm_values, base_poly and offset must be built for the model's max_depth");
    py::array_t<float> out({n, n_num + n_cat, static_cast<py::ssize_t>(md.output_dim)});
    float *dst = out.mutable_data();
    std::fill(dst, dst + out.size(), 0.0f);
    int rc;
    {
        py::gil_scoped_release release;
        rc = whole_ensemble
                 ? gbrl_hip_ensemble_shap(self.h, static_cast<const float *>(o.ptr), static_cast<const char *>(c.ptr), static_cast<int>(n),
                                          static_cast<const float *>(nv.ptr), static_cast<const float *>(bp.ptr), static_cast<const float *>(of.ptr), dst)
                 : gbrl_hip_tree_shap(self.h, tree_idx, static_cast<const float *>(o.ptr), static_cast<const char *>(c.ptr), static_cast<int>(n),
                                      static_cast<const float *>(nv.ptr), static_cast<const float *>(bp.ptr), static_cast<const float *>(of.ptr), dst);
    }
    check(rc);
    return out;
}

}  // namespace

PYBIND11_MODULE(gbrl_cpp, m) {
    m.doc() = "MI355X-native drop-in for NVlabs/gbrl's gbrl_cpp (step / predict hot path), backed by libgbrl_hip.so";
    py::class_<PyGBRL> g(m, "GBRL");
    g.def(py::init<int, int, int, int, int, int, int, float, std::string, std::string, bool, int, std::string, int, std::string, std::string>(),
          py::arg("input_dim") = 1, py::arg("output_dim") = 1, py::arg("policy_dim") = 1, py::arg("max_depth") = 4,
          py::arg("min_data_in_leaf") = 0, py::arg("n_bins") = 256, py::arg("par_th") = 10, py::arg("cv_beta") = 0.9,
          py::arg("split_score_func") = "cosine", py::arg("generator_type") = "quantile",
          py::arg("use_control_variates") = false, py::arg("batch_size") = 5000, py::arg("grow_policy") = "greedy",
          py::arg("verbose") = 0, py::arg("device") = "cpu", py::arg("learner_name") = "GBRL");
    g.def(py::init<const PyGBRL &>(), py::arg("model"));
    g.def_static("load", [](const std::string &filename) {
        gbrl_hip_model *h = gbrl_hip_load(filename.c_str());
        if (!h) fail(gbrl_hip_last_error());
        return new PyGBRL(h);
    }, py::return_value_policy::take_ownership);
    g.def("to_device", [](PyGBRL &self, const std::string &d) { self.device = parse_device(d); }, py::arg("device"));
    g.def("step", &step_impl, py::arg("obs"), py::arg("categorical_obs"), py::arg("grads"));
    g.def("predict", [](PyGBRL &self, py::object &obs, py::object &cat, py::object start, py::object stop, bool return_torch) {
        return predict_impl(self, obs, cat, start, stop, return_torch);
    }, py::arg("obs"), py::arg("categorical_obs"), py::arg("start_tree_idx") = 0, py::arg("stop_tree_idx") = 0, py::arg("return_torch") = false);
    // extension (no counterpart in the reference): categorical cells encoded once, predicted many times (include/gbrl_hip.h)
    g.def("encode_categorical", &encode_categorical_impl, py::arg("categorical_obs"));
    g.def("predict_encoded", [](PyGBRL &self, py::object &obs, py::object &ids, uint64_t token, py::object start, py::object stop, bool return_torch) {
        return predict_impl(self, obs, ids, start, stop, return_torch, &token);
    }, py::arg("obs"), py::arg("categorical_ids"), py::arg("dictionary_token"), py::arg("start_tree_idx") = 0, py::arg("stop_tree_idx") = 0,
          py::arg("return_torch") = false);
    g.def("fit", &fit_impl, py::arg("obs"), py::arg("categorical_obs"), py::arg("targets"), py::arg("iterations"),
          py::arg("shuffle") = true, py::arg("loss_type") = "MultiRMSE");
    g.def("set_bias", [](PyGBRL &self, py::object &bias) {
        const gbrl_hip_metadata md = self.meta();
        Input b = read_input(bias, "bias", false, "set_bias", false);
        int n, dim;  // binding.cpp:622-657
        if (b.shape.size() == 1) {
            if (md.output_dim > 1) { n = 1; dim = static_cast<int>(b.shape[0]); } else { n = static_cast<int>(b.shape[0]); dim = 1; }
        } else {
            n = static_cast<int>(b.shape[0]); dim = static_cast<int>(b.shape[1]);
            if (n == md.output_dim && dim == 1) { n = 1; dim = static_cast<int>(b.shape[0]); }
        }
        if (dim != md.output_dim) {
            std::stringstream ss;
            ss << "Targets output dim " << dim << " != correct output dim " << md.output_dim;
            fail(ss.str());
        }
        if (n > 1) fail("Set bias with multiple samples is not supported!");
        check(gbrl_hip_set_bias(self.h, static_cast<const float *>(b.ptr), md.output_dim, b.on_device));
    });
    g.def("set_feature_weights", [](PyGBRL &self, py::object &w) {
        const gbrl_hip_metadata md = self.meta();
        Input x = read_input(w, "feature_weights", false, "set_feature_weights", false);
        size_t tot = 1;
        for (size_t d : x.shape) tot *= d;
        if (static_cast<int>(tot) != md.input_dim) {
            std::stringstream ss;
            ss << "feature_weights input dim " << tot << " != correct input dim " << md.input_dim;
            fail(ss.str());
        }
        check(gbrl_hip_set_feature_weights(self.h, static_cast<const float *>(x.ptr), md.input_dim, x.on_device));
    });
    g.def("set_feature_mapping", [](PyGBRL &self, const py::array_t<int> &fm, const py::array_t<bool> &mn) {
        if (!(fm.flags() & py::array::c_style) || !(mn.flags() & py::array::c_style)) fail("Arrays must be C-contiguous");
        if (fm.size() != mn.size()) fail("feature_mapping and mapping_numerics must have the same length");
        check(gbrl_hip_set_feature_mapping(self.h, fm.data(), reinterpret_cast<const uint8_t *>(mn.data()), static_cast<int>(fm.size())));
    });
    g.def("get_bias", [](PyGBRL &self) {
        py::array_t<float> a(self.meta().output_dim);
        gbrl_hip_get_bias(self.h, a.mutable_data());
        return a;
    });
    g.def("get_feature_weights", [](PyGBRL &self) {
        py::array_t<float> a(self.meta().input_dim);
        gbrl_hip_get_feature_weights(self.h, a.mutable_data());
        return a;
    });
    g.def("get_feature_mapping", [](PyGBRL &self) {
        const int in = self.meta().input_dim;
        py::array_t<int> fm(in);
        py::array_t<bool> mn(in);
        gbrl_hip_get_feature_mapping(self.h, fm.mutable_data(), reinterpret_cast<uint8_t *>(mn.mutable_data()));
        return py::make_tuple(fm, mn);
    });
    g.def("get_optimizers", [](PyGBRL &self) {
        py::list out;
        for (int i = 0; i < gbrl_hip_num_optimizers(self.h); ++i) {
            gbrl_hip_optimizer o{};
            gbrl_hip_get_optimizer(self.h, i, &o);
            py::dict d;  // keys as binding.cpp:393-410 (including the reference's "eps]" key)
            d["algo"] = o.algo == GBRL_HIP_ALGO_SGD ? "SGD" : "Adam";
            d["init_lr"] = o.init_lr; d["start_idx"] = o.start_idx; d["stop_idx"] = o.stop_idx;
            d["scheduler_func"] = o.scheduler == GBRL_HIP_SCHED_CONST ? "Const" : "Linear";
            d["stop_lr"] = o.stop_lr; d["T"] = o.T; d["beta_1"] = o.beta_1; d["beta_2"] = o.beta_2; d["eps]"] = o.eps;
            out.append(d);
        }
        return out;
    });
    g.def("set_optimizer", [](PyGBRL &self, const std::string &algo, const std::string &sched, float init_lr, int start_idx,
                              int stop_idx, float stop_lr, int T, float beta_1, float beta_2, float eps, float) {
        gbrl_hip_optimizer o{};
        o.algo = parse_enum(algo, {{"SGD", 0}, {"sgd", 0}, {"Adam", 1}, {"adam", 1}}, "Invalid optimizer algorithm! Options are: SGD/Adam");
        o.scheduler = parse_enum(sched, {{"Const", 0}, {"const", 0}, {"Linear", 1}, {"linear", 1}}, "Invalid scheduler! Options are: Const/Linear");
        o.init_lr = init_lr; o.start_idx = start_idx; o.stop_idx = stop_idx; o.stop_lr = stop_lr; o.T = T;
        o.beta_1 = beta_1; o.beta_2 = beta_2; o.eps = eps;
        check(gbrl_hip_set_optimizer(self.h, &o));
    }, py::arg("algo") = "SGD", py::arg("scheduler") = "const", py::arg("init_lr") = 1.0, py::arg("start_idx") = 0,
       py::arg("stop_idx") = 0, py::arg("stop_lr") = 1.0e-8, py::arg("T") = 10000, py::arg("beta_1") = 0.9,
       py::arg("beta_2") = 0.999, py::arg("eps") = 1.0e-8, py::arg("shrinkage") = 0.0);
    g.def("save", [](PyGBRL &self, const std::string &filename) -> int {
        py::gil_scoped_release release;
        const int rc = gbrl_hip_save(self.h, filename.c_str());
        if (rc != GBRL_HIP_OK) { py::gil_scoped_acquire a; fail(gbrl_hip_last_error()); }
        return 0;
    });
    g.def("export", [](PyGBRL &self, const std::string &filename, const std::string &modelname, const std::string &export_format,
                       const std::string &export_type, const std::string &prefix) -> int {
        int rc;
        {
            py::gil_scoped_release release;
            rc = gbrl_hip_export(self.h, filename.c_str(), modelname.c_str(), export_format.c_str(), export_type.c_str(), prefix.c_str());
        }
        check(rc);
        return 0;
    }, py::arg("filename"), py::arg("modelname") = "", py::arg("export_format") = "float", py::arg("export_type") = "full", py::arg("prefix") = "");
    g.def("get_scheduler_lrs", [](PyGBRL &self) {
        const int n = gbrl_hip_num_optimizers(self.h);
        if (n == 0) fail("No optimizers found");
        py::array_t<float> a(n);
        for (int i = 0; i < n; ++i) { gbrl_hip_optimizer o{}; gbrl_hip_get_optimizer(self.h, i, &o); a.mutable_data()[i] = o.init_lr; }
        return a;
    });
    g.def("get_num_trees", [](PyGBRL &self) { return self.meta().n_trees; });
    g.def("get_iteration", [](PyGBRL &self) { return self.meta().iteration; });
    g.def("get_metadata", [](PyGBRL &self) {
        const gbrl_hip_metadata md = self.meta();
        py::dict d;  // keys as binding.cpp:309-328
        d["input_dim"] = md.input_dim; d["output_dim"] = md.output_dim; d["policy_dim"] = md.policy_dim;
        d["split_score_func"] = md.split_score_func == GBRL_HIP_SCORE_L2 ? "L2" : "Cosine";
        d["generator_type"] = md.generator_type == GBRL_HIP_GEN_UNIFORM ? "Uniform" : "Quantile";
        d["use_control_variates"] = md.use_cv != 0; d["verbose"] = md.verbose; d["max_depth"] = md.max_depth;
        d["min_data_in_leaf"] = md.min_data_in_leaf; d["n_bins"] = md.n_bins; d["par_th"] = md.par_th;
        d["batch_size"] = md.batch_size;
        d["grow_policy"] = md.grow_policy == GBRL_HIP_GROW_GREEDY ? "Greedy" : "Oblivious";
        d["iteration"] = md.iteration;
        return d;
    });
    g.def("get_ensemble_data", [](PyGBRL &self) {
        const gbrl_hip_metadata md = self.meta();
        const py::ssize_t T = md.n_trees, L = md.n_leaves, S = md.grow_policy == GBRL_HIP_GROW_OBLIVIOUS ? T : L;
        const py::ssize_t MD = md.max_depth, D = md.output_dim, in = md.input_dim;
        py::array_t<int> tree_indices(T), depths(S), fidx({S, MD}), rn(in), rc(in), fm(in);
        py::array_t<float> values({L, D}), fval({S, MD}), ew({L, MD}), bias(D), fw(in);
        py::array_t<bool> isnum({S, MD}), ineq({L, MD}), mn(in);
        py::array cats(py::dtype("S128"), std::vector<py::ssize_t>{S, MD});
        gbrl_hip_get_ensemble(self.h, tree_indices.mutable_data(), depths.mutable_data(), values.mutable_data(), fidx.mutable_data(),
                              fval.mutable_data(), ew.mutable_data(), reinterpret_cast<uint8_t *>(isnum.mutable_data()),
                              reinterpret_cast<uint8_t *>(ineq.mutable_data()), static_cast<char *>(cats.mutable_data()),
                              rn.mutable_data(), rc.mutable_data());
        gbrl_hip_get_bias(self.h, bias.mutable_data());
        gbrl_hip_get_feature_weights(self.h, fw.mutable_data());
        gbrl_hip_get_feature_mapping(self.h, fm.mutable_data(), reinterpret_cast<uint8_t *>(mn.mutable_data()));
        py::dict d;  // keys as binding.cpp:330-390
        d["bias"] = bias; d["feature_mapping"] = fm; d["reverse_num_feature_mapping"] = rn; d["reverse_cat_feature_mapping"] = rc;
        d["feature_weights"] = fw; d["tree_indices"] = tree_indices; d["depths"] = depths; d["values"] = values;
        d["feature_indices"] = fidx; d["feature_values"] = fval; d["edge_weights"] = ew; d["is_numerics"] = isnum;
        d["inequality_directions"] = ineq; d["mapping_numerics"] = mn; d["categorical_values"] = cats;
        d["alloc_data_size"] = gbrl_hip_alloc_data_size(self.h);
        return d;
    });
    g.def("get_device", [](PyGBRL &self) { return std::string(self.device ? "cuda" : "cpu"); });
    g.def("get_learner_name", [](PyGBRL &self) { return std::string(gbrl_hip_learner_name(self.h)); });
    g.def("print_tree", [](PyGBRL &self, int tree_idx) {
        int rc;
        { py::gil_scoped_release release; rc = gbrl_hip_print_tree(self.h, tree_idx); }
        check(rc);
    }, py::arg("tree_idx") = -1);
    g.def("tree_shap", [](PyGBRL &self, int tree_idx, py::object &obs, py::object &categorical_obs, py::object &norm_values,
                          py::object &base_poly, py::object &offset) {
        return shap_impl(self, false, tree_idx, obs, categorical_obs, norm_values, base_poly, offset);
    }, py::arg("tree_idx") = 0, py::arg("obs"), py::arg("categorical_obs"), py::arg("norm_values"), py::arg("base_poly"), py::arg("offset"));
    g.def("ensemble_shap", [](PyGBRL &self, py::object &obs, py::object &categorical_obs, py::object &norm_values, py::object &base_poly,
                              py::object &offset) {
        return shap_impl(self, true, 0, obs, categorical_obs, norm_values, base_poly, offset);
    }, py::arg("obs"), py::arg("categorical_obs"), py::arg("norm_values"), py::arg("base_poly"), py::arg("offset"));
    g.def("plot_tree", [](PyGBRL &self, int tree_idx, const std::string &filename) {
        check(gbrl_hip_plot_tree(self.h, tree_idx, filename.c_str()));
    }, py::arg("tree_idx") = -1, py::arg("filename"));
    g.def("print_ensemble_metadata", [](PyGBRL &self) {
        int rc;
        { py::gil_scoped_release release; rc = gbrl_hip_print_ensemble_metadata(self.h, self.device ? "cuda" : "cpu"); }
        check(rc);
    });
    // GBRL::cuda_available (gbrl.cpp:542-548): here "is a HIP device usable"
    g.def_static("cuda_available", []() { return gbrl_hip_device_count() > 0; });
    // ---- additions (not in the reference) ----
    g.def("set_profiling", [](PyGBRL &self, int level) { check(gbrl_hip_set_profiling(self.h, level)); });   // True == 1
    g.def("last_phase_times", [](PyGBRL &self) {
        const char *names[64];
        float ms[64];
        const int n = gbrl_hip_last_phase_times(self.h, names, ms, 64);
        py::dict d;
        for (int i = 0; i < n && i < 64; ++i) d[names[i]] = ms[i];
        return d;
    });
    g.def("_handle", [](PyGBRL &self) { return reinterpret_cast<uintptr_t>(self.h); },
          "address of the underlying gbrl_hip_model (for ctypes callers, e.g. gbrl_hip_set_collective)");
}
