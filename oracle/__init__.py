"""oracle -- CPU checkers for the HIP product.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``gbrl_amd``) never does.

Two checkers live here:

* :class:`OracleGBRL` -- ctypes front-end of ``liboracle.so`` (this repo's own restatement,
  ``oracle/oracle.cpp``), with the call surface of the reference's ``gbrl_cpp.GBRL`` for the hot
  path (``step`` / ``predict`` / ``get_ensemble_data`` / setters; ``binding.cpp:421-960``).
* :func:`load_ref` -- imports ``oracle/_ref/gbrl_cpp_ref*.so``, i.e. the reference's own CPU path compiled
  from ``/root/reference`` by ``oracle/Makefile`` (present in the authoring container; travels to the
  GPU box as a prebuilt binary; never committed).
"""
from __future__ import annotations

import ctypes as C
import glob
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_SCORE = {"l2": 0, "cosine": 1}
_GEN = {"uniform": 0, "quantile": 1}
_POLICY = {"greedy": 0, "oblivious": 1}


def build(verbose: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout, out.stderr)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed")


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    fp, ip, u8p, cp = (C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_char_p)
    lib.oracle_create.restype = C.c_void_p
    lib.oracle_create.argtypes = [C.c_int] * 9
    lib.oracle_destroy.argtypes = [C.c_void_p]
    lib.oracle_set_bias.argtypes = [C.c_void_p, fp]
    lib.oracle_set_feature_weights.argtypes = [C.c_void_p, fp]
    lib.oracle_set_feature_mapping.argtypes = [C.c_void_p, ip, u8p]
    lib.oracle_add_sgd.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
    lib.oracle_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.oracle_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p]
    lib.oracle_sizes.argtypes = [C.c_void_p, ip]
    lib.oracle_get_ensemble.argtypes = [C.c_void_p] + [C.c_void_p] * 9
    lib.oracle_last_candidates.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    lib.oracle_last_candidates.restype = C.c_int
    _LIB = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleGBRL:
    """Restated reference CPU path with the gbrl_cpp.GBRL hot-path call surface."""

    def __init__(self, input_dim=1, output_dim=1, policy_dim=1, max_depth=4, min_data_in_leaf=0,
                 n_bins=256, par_th=10, cv_beta=0.9, split_score_func="cosine",
                 generator_type="quantile", use_control_variates=False, batch_size=5000,
                 grow_policy="greedy", verbose=0, device="cpu", learner_name="oracle"):
        if use_control_variates:
            raise RuntimeError("control variates are out of scope (SURVEY.md row 19)")
        self.input_dim, self.output_dim, self.max_depth, self.n_bins = input_dim, output_dim, max_depth, n_bins
        self.grow_policy = grow_policy.lower()
        self._lib = _lib()
        self._h = self._lib.oracle_create(input_dim, output_dim, max_depth, min_data_in_leaf, n_bins,
                                          par_th, _SCORE[split_score_func.lower()],
                                          _GEN[generator_type.lower()], _POLICY[self.grow_policy])

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.oracle_destroy(self._h)
            self._h = None

    def set_bias(self, bias):
        b = np.ascontiguousarray(bias, np.float32).reshape(-1)
        self._lib.oracle_set_bias(self._h, b.ctypes.data_as(C.POINTER(C.c_float)))

    def set_feature_weights(self, w):
        w = np.ascontiguousarray(w, np.float32).reshape(-1)
        self._lib.oracle_set_feature_weights(self._h, w.ctypes.data_as(C.POINTER(C.c_float)))

    def set_feature_mapping(self, mapping, is_numeric):
        m = np.ascontiguousarray(mapping, np.int32)
        n = np.ascontiguousarray(is_numeric, np.uint8)
        self._lib.oracle_set_feature_mapping(self._h, m.ctypes.data_as(C.POINTER(C.c_int32)),
                                             n.ctypes.data_as(C.POINTER(C.c_uint8)))

    def set_optimizer(self, algo="SGD", scheduler="const", init_lr=1.0, start_idx=0, stop_idx=0, **_):
        if algo.lower() != "sgd" or scheduler.lower() != "const":
            raise RuntimeError("oracle restates SGD + Const only")
        if self._lib.oracle_add_sgd(self._h, init_lr, start_idx, stop_idx) != 0:
            raise RuntimeError("invalid optimizer")

    @staticmethod
    def _prep(obs, cat):
        o = None if obs is None else np.ascontiguousarray(obs, np.float32)
        c = None if cat is None else np.ascontiguousarray(cat, "S128")
        if o is not None and o.ndim == 1:
            o = o.reshape(-1, 1)
        if c is not None and c.ndim == 1:
            c = c.reshape(-1, 1)
        return o, c

    def step(self, obs, categorical_obs, grads):
        o, c = self._prep(obs, categorical_obs)
        g = np.ascontiguousarray(grads, np.float32).reshape(-1, self.output_dim)
        n = g.shape[0]
        rc = self._lib.oracle_step(self._h, _ptr(o), _ptr(c), _ptr(g), n,
                                   0 if o is None else o.shape[1], 0 if c is None else c.shape[1])
        if rc != 0:
            raise RuntimeError(f"oracle_step failed ({rc})")

    def predict(self, obs, categorical_obs, start_tree_idx=0, stop_tree_idx=0, return_torch=False):
        o, c = self._prep(obs, categorical_obs)
        n = (o if o is not None else c).shape[0]
        out = np.empty((n, self.output_dim), np.float32)
        rc = self._lib.oracle_predict(self._h, _ptr(o), _ptr(c), n, 0 if o is None else o.shape[1],
                                      0 if c is None else c.shape[1], start_tree_idx or 0,
                                      stop_tree_idx or 0, _ptr(out))
        if rc != 0:
            raise RuntimeError(f"oracle_predict failed ({rc})")
        return out[:, 0].copy() if self.output_dim == 1 else out

    def _sizes(self):
        s = np.zeros(8, np.int32)
        self._lib.oracle_sizes(self._h, s.ctypes.data_as(C.POINTER(C.c_int32)))
        return s

    def get_num_trees(self):
        return int(self._sizes()[0])

    def get_iteration(self):
        return int(self._sizes()[5])

    def get_ensemble_data(self):
        T, L, S, md, D = (int(v) for v in self._sizes()[:5])
        e = {
            "tree_indices": np.zeros(T, np.int32), "depths": np.zeros(S, np.int32),
            "values": np.zeros((L, D), np.float32), "feature_indices": np.zeros((S, md), np.int32),
            "feature_values": np.zeros((S, md), np.float32), "edge_weights": np.zeros((L, md), np.float32),
            "is_numerics": np.zeros((S, md), np.bool_), "inequality_directions": np.zeros((L, md), np.bool_),
            "categorical_values": np.zeros((S, md), "S128"),
        }
        self._lib.oracle_get_ensemble(self._h, *[_ptr(e[k]) for k in (
            "tree_indices", "depths", "values", "feature_indices", "feature_values", "edge_weights",
            "is_numerics", "inequality_directions", "categorical_values")])
        return e

    def last_candidates(self):
        cap = self.n_bins * self.input_dim
        fi, v = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        ic, cat = np.zeros(cap, np.uint8), np.zeros(cap, "S128")
        n = self._lib.oracle_last_candidates(self._h, _ptr(fi), _ptr(v), _ptr(ic), _ptr(cat))
        return fi[:n], v[:n], ic[:n].astype(bool), cat[:n]


def ref_path(native: bool = False):
    pat = os.path.join(_HERE, "_ref", "native" if native else "", "gbrl_cpp_ref*.so")
    hits = sorted(glob.glob(pat))
    return hits[0] if hits else None


def load_ref_capacity():
    """The capacity-only patched reference build (oracle/Makefile target ref-capacity: INITAL_MAX_TREES 50000 -> 16384 in a
    /tmp copy, so that greedy models with max_depth >= 6 can be constructed).  Returns the module or None."""
    hits = sorted(glob.glob(os.path.join(_HERE, "_ref", "capacity", "gbrl_cpp_refcap*.so")))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("gbrl_cpp_refcap", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref(native: bool = False):
    """Import the reference's own CPU build (oracle/_ref).  Returns the module or None."""
    path = ref_path(native)
    if path is None:
        return None
    spec = importlib.util.spec_from_file_location("gbrl_cpp_ref", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- the reference's per-candidate float32 score as an explicit operation sequence (replay_sequence.cpp) and the reference's own
# ---- functions behind a C shim (oracle/_ref/ref_score_probe.so): what the product's near-tie replay is tested against
def replay_scores(obs, grads, rows, feature, value, min_data, cosine):
    """(split score, parent score) of the numeric candidate obs[:, feature] > value on the node `rows` (ascending), restated sequence."""
    lib = _lib()
    obs = np.ascontiguousarray(obs, np.float32)
    grads = np.ascontiguousarray(grads, np.float32)
    rows = np.ascontiguousarray(rows, np.int32)
    n, F, D = len(rows), obs.shape[1], grads.shape[1]
    sp = lib.oracle_replay_split_cosine if cosine else lib.oracle_replay_split_l2
    pa = lib.oracle_replay_parent_cosine if cosine else lib.oracle_replay_parent_l2
    sp.restype = pa.restype = C.c_float
    sp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    pa.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    s = sp(obs.ctypes.data, grads.ctypes.data, rows.ctypes.data, n, F, D, int(feature), float(value), int(min_data))
    p = pa(grads.ctypes.data, rows.ctypes.data, n, D) if n else 0.0
    return np.float32(s), np.float32(p)


_PROBE = None


def ref_score_probe():
    """ctypes handle onto TreeNode::splitScoreCosine / splitScoreL2 / scoreCosine / scoreL2 of oracle/_ref, or None when it is not built."""
    global _PROBE
    if _PROBE is None:
        path = os.path.join(_HERE, "_ref", "ref_score_probe.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.ref_split_score.restype = lib.ref_parent_score.restype = C.c_float
        lib.ref_split_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        lib.ref_parent_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.ref_split_score_cat.restype = C.c_float
        lib.ref_split_score_cat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
        _PROBE = lib
    return _PROBE


def ref_scores(obs, grads, rows, feature, value, min_data, cosine):
    """The same pair from the reference's own functions."""
    lib = ref_score_probe()
    obs = np.ascontiguousarray(obs, np.float32)
    grads = np.ascontiguousarray(grads, np.float32)
    rows = np.ascontiguousarray(rows, np.int32)
    n, F, D = len(rows), obs.shape[1], grads.shape[1]
    s = lib.ref_split_score(obs.ctypes.data, grads.ctypes.data, rows.ctypes.data, n, F, D, int(feature), float(value), int(min_data), 1 if cosine else 0)
    p = lib.ref_parent_score(grads.ctypes.data, rows.ctypes.data, n, D, 1 if cosine else 0) if n else 0.0
    return np.float32(s), np.float32(p)


def ref_split_score_cat(cat_obs, grads, rows, feature, value, min_data, cosine):
    """TreeNode::splitScoreCosineCategorical / splitScoreL2Categorical of the reference build: candidate cat_obs[:, feature] == value."""
    lib = ref_score_probe()
    cat_obs = np.ascontiguousarray(cat_obs, "S128")
    grads = np.ascontiguousarray(grads, np.float32)
    rows = np.ascontiguousarray(rows, np.int32)
    return np.float32(lib.ref_split_score_cat(cat_obs.ctypes.data, grads.ctypes.data, rows.ctypes.data, len(rows), cat_obs.shape[1], grads.shape[1], int(feature),
                                              bytes(value).ljust(128, b"\0"), int(min_data), 1 if cosine else 0))
