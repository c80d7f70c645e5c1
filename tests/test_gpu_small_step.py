"""RL-sized steps (round 4): the fused kernels that cut their launch count must not change a bit.

* `k_small_stats` (one launch for column sums, mean, centred squares, scales and quantisation) replays the reduction tree of the
  multi-kernel chain on virtual threads: the quantised gradients, and therefore every tree, are the same (`GBRL_HIP_NO_SMALL_STATS=1`
  is the chain).
* `k_sort_quantiles` writes the class codes of its feature itself (`GBRL_HIP_SORT_NO_CODES=1` = the separate binning kernel).
* `k_hist_build` stores the int64 histograms of single-chunk nodes itself (`GBRL_HIP_NO_DIRECT_HIST=1` = partials + `k_hist_reduce`).
The hooks are read per call (a latched hook would compare a path with itself).  All against the plain paths on batches of 2 .. 4096 rows, both policies and scores, 1 .. 16 outputs, feature counts with and without a
partial last code group, categorical columns beside the numeric ones -- ensembles compared byte for byte.
"""
import numpy as np
import pytest

import cases as K

pytestmark = pytest.mark.gpu

HOOKS = ("GBRL_HIP_NO_SMALL_STATS", "GBRL_HIP_SORT_NO_CODES", "GBRL_HIP_NO_DIRECT_HIST")


def _grow(case, monkeypatch, env):
    import gbrl_amd
    for k in HOOKS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    pred = K.drive(m, case, X, Xc, G, y)
    return m.get_ensemble_data(), np.asarray(pred)


@pytest.mark.parametrize("N", [2, 37, 256, 1000, 4096])
@pytest.mark.parametrize("policy,score,gen,D,F,Fc", [("greedy", "L2", "Quantile", 1, 16, 0), ("oblivious", "L2", "Quantile", 8, 24, 0),
                                                    ("greedy", "Cosine", "Quantile", 3, 5, 2), ("oblivious", "Cosine", "Uniform", 16, 33, 0),
                                                    ("oblivious", "L2", "Quantile", 5, 40, 3), ("greedy", "L2", "Uniform", 8, 130, 1)])
def test_fused_small_step_kernels_keep_every_bit(policy, score, gen, D, F, Fc, N, monkeypatch):
    case = dict(name="ss", seed=40 + N + D, N=N, F=F, Fc=Fc, D=D, depth=5 if F > 100 else 4, n_bins=64, score=score, gen=gen, policy=policy, trees=4, min_data_in_leaf=2 if D == 3 else 0,
                loop="rmse" if D == 1 else None)
    if case["loop"] is None:
        del case["loop"]
    ref, pref = _grow(case, monkeypatch, {k: "1" for k in HOOKS})
    for env in ({}, {"GBRL_HIP_NO_SMALL_STATS": "1", "GBRL_HIP_NO_DIRECT_HIST": "1"}, {"GBRL_HIP_SORT_NO_CODES": "1"}, {"GBRL_HIP_NO_DIRECT_HIST": "1"}):
        got, pgot = _grow(case, monkeypatch, env)
        for k in ref:
            a, b = np.asarray(ref[k]), np.asarray(got[k])
            assert a.shape == b.shape and a.tobytes() == b.tobytes(), (env, k)
        assert pref.tobytes() == pgot.tobytes(), env


def test_the_hooks_really_switch_paths(monkeypatch):
    """Phase table of one step with and without the fused kernels: only the plain path has a `hist_reduce` phase."""
    import gbrl_amd
    case = dict(name="sw", seed=7, N=1024, F=24, Fc=0, D=4, depth=4, n_bins=64, score="L2", gen="Quantile", policy="oblivious", trees=1)
    X, Xc, G, y = K.make_inputs(case)
    def launches(env):
        for k in HOOKS:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        m.set_profiling(2)
        K.drive(m, case, X, Xc, G, y)
        m.step(X, Xc, np.ascontiguousarray(G))     # (drive ends with a predict: the phase table is the last call's)
        return dict(m.last_phase_times())
    fused, plain = launches({}), launches({k: "1" for k in HOOKS})
    # 1024 rows: every node is one chunk, so the fused path never launches k_hist_reduce (and records no such phase)
    assert "hist_reduce" in plain and "hist_reduce" not in fused, (plain, fused)
