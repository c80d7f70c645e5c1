"""`k_predict_grd_stream` (predict_grd_stream.hip, round 6): small greedy ensembles over large batches -- the configs[2]
half of the headline metric -- take a barrier-free kernel with the whole ensemble in LDS and one row-tile pipeline per wave.  Its bits must be
those of the block-cooperative kernel (`GBRL_HIP_PREDICT_NO_GRD_STREAM=1`: k_predict_obl2<GREEDY>) and of the general kernel
(`GBRL_HIP_PREDICT_GENERIC=1`: the reference's leaf-by-leaf walk, predictor.cpp:188-229, with optimizer.cpp:110-118's update per tree):
the same fma chain per output in tree order.  Shapes: 32 / 64 / 96 / 128 features, 1..8 outputs, depth 3..6, as many trees as fit in LDS
beside the row tiles and one more (which must fall back), full tiles and a ragged last one, tree ranges, NaN / inf cells.  The kernel takes
batches of 2^18 rows or more by default (below that it has too few tiles per wave to win); `GBRL_HIP_PREDICT_GRD_STREAM_MIN_ROWS=1` lets these
tests run it on 33 000 .. 45 000 rows.
"""
import numpy as np
import pytest

import cases as K

pytestmark = pytest.mark.gpu

HOOKS = ("GBRL_HIP_PREDICT_NO_GRD_STREAM", "GBRL_HIP_PREDICT_GENERIC")


def _model(depth, D, F, trees, seed, score="Cosine"):
    import gbrl_amd
    case = dict(name="gs", seed=seed, N=3000, F=F, Fc=0, D=D, depth=depth, n_bins=64, score=score, gen="Quantile", policy="greedy", trees=trees)
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, Xc, G, y)
    assert m.get_num_trees() == trees
    return m


def _predict_all(m, X, monkeypatch, start=0, stop=0):
    monkeypatch.setenv("GBRL_HIP_PREDICT_GRD_STREAM_MIN_ROWS", "1")      # (by default the kernel takes batches of 2^18+ rows)
    outs = {}
    for mode, env in (("default", {}), ("cooperative", {"GBRL_HIP_PREDICT_NO_GRD_STREAM": "1"}), ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"})):
        for h in HOOKS:
            monkeypatch.delenv(h, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[mode] = np.asarray(m.predict(X, None, start, stop)).copy()
    for h in HOOKS:
        monkeypatch.delenv(h, raising=False)
    return outs


def _same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("F", [32, 64, 96, 128])
@pytest.mark.parametrize("D,depth,trees", [(8, 6, 10), (8, 6, 3), (5, 6, 7), (8, 4, 9), (3, 5, 6), (1, 6, 10), (4, 3, 12), (2, 6, 1), (8, 6, 17)])
def test_streaming_greedy_predict_is_bitwise_the_other_kernels(F, D, depth, trees, monkeypatch):
    m = _model(depth, D, F, trees, seed=8100 + F + 7 * D + depth)
    rng = np.random.default_rng(F * 1000 + D * 10 + depth)
    n = 32768 + 64 * int(rng.integers(0, 200)) + int(rng.integers(1, 64))      # full tiles and a ragged last one
    X = rng.standard_normal((n, F)).astype(np.float32)
    X[rng.integers(0, n, 50), rng.integers(0, F, 50)] = np.nan                  # x > t is false for NaN on every path
    X[rng.integers(0, n, 50), rng.integers(0, F, 50)] = np.inf
    X[rng.integers(0, n, 50), rng.integers(0, F, 50)] = -np.inf
    outs = _predict_all(m, X, monkeypatch)
    assert _same(outs["default"], outs["cooperative"]), "streaming kernel differs from k_predict_obl2<GREEDY>"
    assert _same(outs["default"], outs["generic"]), "streaming kernel differs from the general kernel"
    if trees >= 3:      # a tree range (start / stop inside the ensemble)
        outs = _predict_all(m, X, monkeypatch, 1, trees - 1)
        assert _same(outs["default"], outs["cooperative"]) and _same(outs["default"], outs["generic"])


def test_the_streaming_kernel_is_the_one_that_runs_and_larger_ensembles_fall_back(monkeypatch):
    """configs[2]'s predict shape (128 features, 8 outputs, depth 6, 10 trees) is taken by k_predict_grd_stream -- checked through the kernel
    time the engine reports with and without the hook (the cooperative kernel is 14 % slower at 2^19 rows, 36 % at 2^20).  Up to 10 trees fit
    beside four row tiles, up to 21 beside three (the three-wave shape); a 22nd no longer fits: the call falls back and still gives the same bits."""
    m = _model(6, 8, 128, 22, seed=8555)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((1 << 19, 128)).astype(np.float32)
    m.set_profiling(1)
    t = {}
    for mode, env in (("default", None), ("cooperative", "1")):
        if env is None:
            monkeypatch.delenv("GBRL_HIP_PREDICT_NO_GRD_STREAM", raising=False)
        else:
            monkeypatch.setenv("GBRL_HIP_PREDICT_NO_GRD_STREAM", env)
        best = 1e9
        for _ in range(6):
            m.predict(X, None, 0, 10)
            best = min(best, m.last_phase_times().get("predict", 1e9))
        t[mode] = best
    monkeypatch.delenv("GBRL_HIP_PREDICT_NO_GRD_STREAM", raising=False)
    print("10 greedy trees, 2^19 x 128 rows: streaming %.1f us, cooperative %.1f us" % (t["default"] * 1e3, t["cooperative"] * 1e3))
    assert t["default"] < 0.97 * t["cooperative"], t
    for start, stop in ((0, 11), (0, 21), (1, 22), (0, 22), (12, 22)):      # three waves (11, 21 trees), not covered (22), four waves from a later record
        outs = _predict_all(m, X[:40000], monkeypatch, start, stop)
        assert _same(outs["default"], outs["cooperative"]) and _same(outs["default"], outs["generic"]), (start, stop)
