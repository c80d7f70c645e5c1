"""GPU tests of the register-tile predict kernel (gbrl_amd/csrc/predict_reg.hip, round 4): oblivious, numeric-only ensembles over
large batches.  The kernel keeps the reference's per-row, per-output chain p = fma(-lr, v, p) in tree order (predictor.cpp:231-265,
optimizer.cpp:110-118), so its outputs must be BITWISE those of the general kernel (`GBRL_HIP_PREDICT_GENERIC=1`, the
walk written like the reference's loop) -- and of the second-generation kernel it replaces for these shapes.

`GBRL_HIP_PREDICT_REG_ONLY=1` makes predict() raise when the register-tile kernel declines a shape, so a test that passes has run
the kernel it names; `GBRL_HIP_PREDICT_REG_MIN_ROWS` lowers the batch size from which it is used (32 768 rows by default) and
`GBRL_HIP_PREDICT_REG_GROUPED` forces its grouped launch shape (k > 1: groups of k trees) on ensembles small enough to stay resident.
"""
import numpy as np
import pytest

import cases as K

pytestmark = pytest.mark.gpu

HOOKS = ("GBRL_HIP_PREDICT_GENERIC", "GBRL_HIP_PREDICT_NO_REG", "GBRL_HIP_PREDICT_REG_ONLY", "GBRL_HIP_PREDICT_REG_MIN_ROWS",
         "GBRL_HIP_PREDICT_REG_GROUPED", "GBRL_HIP_PREDICT_OBL1")


def _set(monkeypatch, env):
    for k in HOOKS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


def _grow(case):
    import gbrl_amd
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, Xc, G, y)
    assert m.get_num_trees() == case["trees"]
    return m, X


def _batch(case, n, seed):
    """n rows of the case's width: the growth rows repeated and perturbed, so that every leaf is visited"""
    rng = np.random.default_rng(seed)
    X, _, _, _ = K.make_inputs(case)
    idx = rng.integers(0, X.shape[0], size=n)
    return np.ascontiguousarray(X[idx] + rng.standard_normal((n, X.shape[1])).astype(np.float32) * np.float32(0.05))


MODES = (("reg", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1"}),
         ("reg_grouped", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "1"}),
         ("reg_groups_of_4", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "4"}),
         ("reg_groups_of_3", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "3"}),
         ("gen2", {"GBRL_HIP_PREDICT_NO_REG": "1"}),
         ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"}))


@pytest.mark.parametrize("depth", [1, 3, 4, 5, 6])
@pytest.mark.parametrize("D,F", [(8, 128), (3, 12), (4, 64), (7, 20), (1, 4), (5, 100)])
def test_register_tile_kernel_equals_the_general_kernel(depth, D, F, monkeypatch):
    """Every compiled variant (4 / 6 levels x 4 / 8 padded outputs), row widths below the register bank (the computed jump of the
    tile load), output counts below the padded width, batches that end inside a 64-row tile, tree ranges that start and stop at odd
    and even trees (the pipelined pair loop + the single-tree tail), both launch shapes and groups that end inside the range."""
    n_trees = 27
    opts = [dict(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=max(1, D - 1))]
    if D > 1:
        opts.append(dict(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=D - 1, stop_idx=D))
    case = dict(name="pr", seed=900 + depth + 7 * D, N=1500, F=F, Fc=0, D=D, depth=depth, n_bins=32, score="L2", gen="Uniform", policy="oblivious",
                trees=n_trees, opts=opts)
    m, _ = _grow(case)
    ranges = ((0, 0), (0, 1), (0, 2), (3, 20), (4, 21), (11, 12), (16, 27), (26, 27))
    for n in (1, 63, 64, 65, 1000, 4133):
        Xp = _batch(case, n, seed=n)
        outs = {}
        for mode, env in MODES:
            _set(monkeypatch, env)
            outs[mode] = [np.asarray(m.predict(Xp, None, a, b)) for a, b in ranges]
        for mode, _ in MODES:
            for r, a, b in zip(ranges, outs[mode], outs["generic"]):
                assert a.shape == b.shape and np.array_equal(a, b), (mode, n, r)
        assert np.abs(outs["reg"][0]).max() > 0


def test_register_tile_kernel_on_a_large_batch_and_a_large_ensemble(monkeypatch):
    """The shapes it exists for: 2^17 rows x 128 features, 8 outputs, depth 6; 30 trees (resident, two blocks per CU), 60 trees
    (resident, one 512-thread block per CU) and 130 trees (grouped: nine groups, the last one partial) -- default dispatch, no
    row-count override -- bitwise equal to the general kernel and to the second-generation kernel."""
    n_trees = 130
    case = dict(name="prl", seed=4242, N=3000, F=128, Fc=0, D=8, depth=6, n_bins=64, score="L2", gen="Quantile", policy="oblivious", trees=n_trees,
                opts=[dict(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=7),
                      dict(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=7, stop_idx=8)])
    m, _ = _grow(case)
    Xp = _batch(case, (1 << 17) + 77, seed=5)
    ranges = ((0, 30), (1, 30), (0, 60), (7, 66), (0, 0), (3, 130), (0, 129))
    outs = {}
    for mode, env in (("reg", {"GBRL_HIP_PREDICT_REG_ONLY": "1"}), ("gen2", {"GBRL_HIP_PREDICT_NO_REG": "1"}), ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"})):
        _set(monkeypatch, env)
        outs[mode] = [np.asarray(m.predict(Xp, None, a, b)) for a, b in ranges]
    for mode in ("reg", "gen2"):
        for r, a, b in zip(ranges, outs[mode], outs["generic"]):
            assert np.array_equal(a, b), (mode, r)


def test_register_tile_kernel_takes_device_inputs_and_declines_what_it_does_not_cover(monkeypatch):
    """Device-resident inputs (the bench's path) through DLPack; a feature count that is not a multiple of four and a categorical
    model are declined (REG_ONLY raises) and predicted by the older kernels as before."""
    import torch
    case = dict(name="prd", seed=77, N=2000, F=24, Fc=0, D=4, depth=4, n_bins=32, score="L2", gen="Uniform", policy="oblivious", trees=9)
    m, _ = _grow(case)
    Xp = _batch(case, 40000, seed=1)
    _set(monkeypatch, {"GBRL_HIP_PREDICT_GENERIC": "1"})
    want = np.asarray(m.predict(Xp, None, 0, 0))
    _set(monkeypatch, {})
    t = torch.from_numpy(Xp).to("cuda:0")
    md = __import__("gbrl_amd").GBRL(**K.ctor_kwargs(case, device="cuda"))
    X, Xc, G, y = K.make_inputs(case)
    K.drive(md, case, X, Xc, G, y)
    _set(monkeypatch, {"GBRL_HIP_PREDICT_REG_ONLY": "1"})
    got = torch.from_dlpack(md.predict((t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda"), None, 0, 0)).cpu().numpy()
    assert np.array_equal(got, want)
    odd = dict(case, F=23, name="prd_odd")
    _set(monkeypatch, {})
    mo, _ = _grow(odd)
    Xo = _batch(odd, 40000, seed=2)
    _set(monkeypatch, {"GBRL_HIP_PREDICT_REG_ONLY": "1"})
    with pytest.raises(RuntimeError, match="REG_ONLY"):
        mo.predict(Xo, None, 0, 0)
    _set(monkeypatch, {})
    a = np.asarray(mo.predict(Xo, None, 0, 0))
    _set(monkeypatch, {"GBRL_HIP_PREDICT_GENERIC": "1"})
    assert np.array_equal(a, np.asarray(mo.predict(Xo, None, 0, 0)))
