"""RL-sized steps (round 4): the fused kernels that cut their launch count must not change a bit.

* `k_small_stats` (one launch for column sums, mean, centred squares, scales and quantisation) replays the reduction tree of the
  multi-kernel chain on virtual threads: the quantised gradients, and therefore every tree, are the same (`GBRL_HIP_NO_SMALL_STATS=1`
  is the chain).
* `k_sort_quantiles` writes the class codes of its feature itself (`GBRL_HIP_SORT_NO_CODES=1` = the separate binning kernel).
* `k_hist_build` stores the int64 histograms of single-chunk nodes itself (`GBRL_HIP_NO_DIRECT_HIST=1` = partials + `k_hist_reduce`).
* round 5: `k_small_grow` grows the WHOLE tree of a step of at most 8192 rows in one launch (LDS histograms per feature slot, one grid
  barrier per level, leaf sums; `small_grow.hip`).  `GBRL_HIP_NO_SMALL_GROW=1` = the level-synchronous loop of kernels.
* round 5: `k_small_prep` runs the whole preparation (statistics, quantisation, quantile or uniform candidates, class codes) as one launch
  from the row-major matrix (`small_prep.hip`; the same device bodies as the separate kernels).  `GBRL_HIP_NO_SMALL_PREP=1` = the separate launches.
The hooks are read per call (a latched hook would compare a path with itself).  All against the plain paths on batches of 2 .. 4096 rows, both policies and scores, 1 .. 16 outputs, feature counts with and without a
partial last code group, categorical columns beside the numeric ones -- ensembles compared byte for byte.
"""
import numpy as np
import pytest

import cases as K

pytestmark = pytest.mark.gpu

HOOKS = ("GBRL_HIP_NO_SMALL_STATS", "GBRL_HIP_SORT_NO_CODES", "GBRL_HIP_NO_DIRECT_HIST", "GBRL_HIP_NO_SMALL_GROW", "GBRL_HIP_NO_SMALL_PREP")


def _grow(case, monkeypatch, env):
    import gbrl_amd
    for k in HOOKS + ("GBRL_HIP_SMALL_GROW_BLOCKS",):
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
    for env in ({}, {"GBRL_HIP_NO_SMALL_STATS": "1", "GBRL_HIP_NO_DIRECT_HIST": "1"}, {"GBRL_HIP_SORT_NO_CODES": "1"}, {"GBRL_HIP_NO_DIRECT_HIST": "1"},
                {"GBRL_HIP_NO_SMALL_GROW": "1"}, {"GBRL_HIP_NO_SMALL_GROW": "1", "GBRL_HIP_NO_DIRECT_HIST": "1"}, {"GBRL_HIP_NO_SMALL_PREP": "1"},
                {"GBRL_HIP_NO_SMALL_PREP": "1", "GBRL_HIP_NO_SMALL_GROW": "1"}, {"GBRL_HIP_NO_SMALL_STATS": "1"}):
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
    # ... and the one-launch growth replaces every level phase
    assert "small_grow" in fused and "small_grow" not in plain and "score_select" in plain and "score_select" not in fused, (plain, fused)
    loop = launches({"GBRL_HIP_NO_SMALL_GROW": "1"})
    assert "small_grow" not in loop and "score_select" in loop and "hist_reduce" not in loop, loop
    # the fused preparation really ran (Engine::last_step_launches counts the preparation's kernel launches: 1 fused, >= 3 separate)
    assert fused["prep_launches"] == 1 and plain["prep_launches"] >= 3, (fused, plain)


def _same_bytes(a, b, what):
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), (what, k)


# (policy, score, generator, outputs, numeric, categorical, n_bins, depth, min_data_in_leaf)
SMALL_GROW_SHAPES = [
    ("greedy", "L2", "Quantile", 1, 16, 0, 256, 4, 0),        # BASELINE configs[0]
    ("oblivious", "L2", "Uniform", 8, 40, 12, 256, 6, 0),     # configs[4]'s minibatch shape, narrower
    ("oblivious", "Cosine", "Quantile", 2, 3, 0, 16, 8, 0),   # deepest tree the kernel takes, empty nodes on the way
    ("greedy", "Cosine", "Quantile", 9, 7, 1, 33, 7, 3),      # two field chunks (D + 1 = 10), min_data rejections, ragged last class tile
    ("greedy", "L2", "Uniform", 17, 20, 0, 100, 5, 0),        # three field chunks
    ("oblivious", "L2", "Quantile", 4, 300, 0, 64, 3, 1),     # more slots than blocks: several slots per block
    ("greedy", "L2", "Quantile", 3, 0, 4, 8, 4, 0),           # categorical only
    ("oblivious", "Cosine", "Quantile", 5, 1, 0, 1, 2, 0),    # one candidate
]


@pytest.mark.parametrize("N", [1, 2, 3, 65, 777, 4096, 4097, 8192])
@pytest.mark.parametrize("shape", SMALL_GROW_SHAPES, ids=lambda t: "-".join(str(x) for x in t))
def test_one_launch_growth_keeps_every_bit(shape, N, monkeypatch):
    """The tree of `k_small_grow` against the level loop's: every array of the ensemble and the predictions, byte for byte.  4097 and 8192
    rows take the int64 LDS accumulators, 1 .. 3 rows mostly empty nodes, depth 8 the largest node table."""
    policy, score, gen, D, F, Fc, B, depth, mdl = shape
    case = dict(name="sg", seed=900 + N + 7 * D + depth, N=N, F=F, Fc=Fc, D=D, depth=depth, n_bins=B, score=score, gen=gen, policy=policy, trees=3,
                min_data_in_leaf=mdl, discrete_cols=[0] if F > 2 else [], constant_cols=[1] if F > 4 else [])
    if F > 0 and D == 1:
        case["loop"] = "rmse"
    loop, ploop = _grow(case, monkeypatch, {"GBRL_HIP_NO_SMALL_GROW": "1", "GBRL_HIP_NO_SMALL_PREP": "1"})
    one, pone = _grow(case, monkeypatch, {})
    _same_bytes(loop, one, "small_grow")
    assert ploop.tobytes() == pone.tobytes()
    sep, psep = _grow(case, monkeypatch, {"GBRL_HIP_NO_SMALL_PREP": "1"})     # the growth kernel behind the separate preparation launches
    _same_bytes(loop, sep, "small_grow, separate preparation")
    # fewer blocks than slots (several slots per block, the codes reloaded per slot) must not matter either
    few, pfew = _grow(case, monkeypatch, {"GBRL_HIP_SMALL_GROW_BLOCKS": "3"})
    _same_bytes(loop, few, "small_grow, 3 blocks")


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES if c["N"] <= 8192 and not c.get("long_loop")])
def test_golden_inputs_through_both_growth_paths(name, monkeypatch):
    """Every golden fixture of at most 8192 rows: the one-launch growth and the level loop give the same ensemble bytes (the fixtures
    themselves are checked against the default path by test_gpu_parity.py)."""
    case = K.BY_NAME[name]
    loop, ploop = _grow(case, monkeypatch, {"GBRL_HIP_NO_SMALL_GROW": "1", "GBRL_HIP_NO_SMALL_PREP": "1"})
    one, pone = _grow(case, monkeypatch, {})
    _same_bytes(loop, one, name)
    assert ploop.tobytes() == pone.tobytes()


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("policy,score,Fc", [("greedy", "Cosine", 0), ("oblivious", "L2", 2)])
def test_a_failed_one_launch_growth_falls_back_to_the_level_loop(policy, score, Fc, mode, monkeypatch):
    """`k_small_grow` assumes its blocks are co-resident; when the launch fails (mode 1) or its blocks abandon a grid
    barrier (mode 2; both simulated by `GBRL_HIP_TEST_SMALL_GROW_FAIL`, read per call) nothing has been booked yet, so the level loop grows
    the tree -- the step must NOT throw -- and the engine keeps to the level loop afterwards (latched: the third tree is grown with the hook
    removed and still never launches the one-launch kernel).  Same bytes as an undisturbed model."""
    import gbrl_amd
    case = dict(name="fb", seed=77, N=1500, F=12, Fc=Fc, D=3, depth=4, n_bins=64, score=score, gen="Quantile", policy=policy, trees=2)
    X, Xc, G, y = K.make_inputs(case)
    for k in HOOKS + ("GBRL_HIP_SMALL_GROW_BLOCKS", "GBRL_HIP_TEST_SMALL_GROW_FAIL"):
        monkeypatch.delenv(k, raising=False)
    ref = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(ref, dict(case, trees=3), X, Xc, G, y)
    monkeypatch.setenv("GBRL_HIP_TEST_SMALL_GROW_FAIL", mode)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    m.set_profiling(2)
    K.drive(m, case, X, Xc, G, y)                      # two trees with the failure injected
    monkeypatch.delenv("GBRL_HIP_TEST_SMALL_GROW_FAIL")
    m.step(X, Xc, np.ascontiguousarray(G))            # a third one without: the latch keeps the level loop
    ph = dict(m.last_phase_times())
    assert ph.get("small_grow_fallbacks", 0) == 1, ph   # one failure, then latched off
    assert "small_grow" not in ph and "score_select" in ph, ph
    a, b = ref.get_ensemble_data(), m.get_ensemble_data()
    for k in a:
        assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), k
