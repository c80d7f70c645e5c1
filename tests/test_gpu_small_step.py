"""RL-sized steps (round 4): the fused kernels that cut their launch count must not change a bit.

* `k_small_stats` (one launch for column sums, mean, centred squares, scales and quantisation) replays the reduction tree of the
  multi-kernel chain on virtual threads: the quantised gradients, and therefore every tree, are the same (`GBRL_HIP_NO_SMALL_STATS=1`
  is the chain).
* `k_sort_quantiles` writes the class codes of its feature itself (`GBRL_HIP_SORT_NO_CODES=1` = the separate binning kernel).
Both against the plain paths on batches of 2 .. 4096 rows, both policies and scores, 1 .. 16 outputs, feature counts with and without a
partial last code group, categorical columns beside the numeric ones -- ensembles compared byte for byte.
"""
import numpy as np
import pytest

import cases as K

pytestmark = pytest.mark.gpu

HOOKS = ("GBRL_HIP_NO_SMALL_STATS", "GBRL_HIP_SORT_NO_CODES")


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
    ref, pref = _grow(case, monkeypatch, {"GBRL_HIP_NO_SMALL_STATS": "1", "GBRL_HIP_SORT_NO_CODES": "1"})
    for env in ({}, {"GBRL_HIP_NO_SMALL_STATS": "1"}, {"GBRL_HIP_SORT_NO_CODES": "1"}):
        got, pgot = _grow(case, monkeypatch, env)
        for k in ref:
            a, b = np.asarray(ref[k]), np.asarray(got[k])
            assert a.shape == b.shape and a.tobytes() == b.tobytes(), (env, k)
        assert pref.tobytes() == pgot.tobytes(), env
