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

HOOKS = ("GBRL_HIP_PREDICT_GENERIC", "GBRL_HIP_PREDICT_NO_REG", "GBRL_HIP_PREDICT_NO_PC", "GBRL_HIP_PREDICT_REG_ONLY", "GBRL_HIP_PREDICT_REG_MIN_ROWS",
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
    return _specials(rng, np.ascontiguousarray(X[idx] + rng.standard_normal((n, X.shape[1])).astype(np.float32) * np.float32(0.05)))


def _specials(rng, X):
    """one cell in forty is a value a threshold comparison must treat like the reference's `x > t`: NaN (never greater), +-inf, +-0,
    and exact copies of other cells (ties with thresholds that are data values)"""
    pick = rng.integers(0, 40, size=X.shape)
    X[pick == 0] = np.float32(np.nan)
    X[pick == 1] = np.float32(np.inf)
    X[pick == 2] = np.float32(-np.inf)
    X[pick == 3] = np.float32(0.0)
    X[pick == 4] = np.float32(-0.0)
    return X


MODES = (("reg", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1"}),
         ("reg_grouped", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "1"}),
         ("reg_groups_of_4", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "4"}),
         ("reg_groups_of_3", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "3"}),
         ("packed", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_NO_REG": "1"}),   # the same rows as packed codes
         ("gen2", {"GBRL_HIP_PREDICT_NO_REG": "1", "GBRL_HIP_PREDICT_NO_PC": "1"}),
         ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"}))


@pytest.mark.parametrize("depth", [1, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("D,F", [(8, 128), (3, 12), (4, 64), (7, 20), (1, 4), (5, 100), (12, 40), (16, 128)])
def test_register_tile_kernel_equals_the_general_kernel(depth, D, F, monkeypatch):
    """Every compiled variant (4 / 6 levels x 4 / 8 / 16 padded outputs; round 5: 8 levels x 4 / 8 padded outputs, the single-record-buffer
    pipeline; the packed-code kernels stop at 6 levels and 8 outputs), row widths below the register bank (the computed jump of the
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
    modes = [mm for mm in MODES if D <= 8 or mm[0] != "packed"]
    if depth > 6:   # 7-8 levels: fp32 register tiles up to 8 outputs, no packed-code variant; wider outputs stay on the older kernels
        modes = [mm for mm in modes if mm[0] != "packed" and (D <= 8 or mm[0] in ("gen2", "generic"))]
    for n in (1, 63, 64, 65, 1000, 4133):
        Xp = _batch(case, n, seed=n)
        outs = {}
        for mode, env in modes:
            _set(monkeypatch, env)
            outs[mode] = [np.asarray(m.predict(Xp, None, a, b)) for a, b in ranges]
        for mode, _ in modes:
            for r, a, b in zip(ranges, outs[mode], outs["generic"]):
                assert a.shape == b.shape and np.array_equal(a, b), (mode, n, r)
        assert np.abs(outs["generic"][0]).max() > 0


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
    for mode, env in (("reg", {"GBRL_HIP_PREDICT_REG_ONLY": "1"}), ("packed", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_NO_REG": "1"}),
                      ("gen2", {"GBRL_HIP_PREDICT_NO_REG": "1", "GBRL_HIP_PREDICT_NO_PC": "1"}), ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"})):
        _set(monkeypatch, env)
        outs[mode] = [np.asarray(m.predict(Xp, None, a, b)) for a, b in ranges]
    for mode in ("reg", "packed", "gen2"):
        for r, a, b in zip(ranges, outs[mode], outs["generic"]):
            assert np.array_equal(a, b), (mode, r)


@pytest.mark.parametrize("depth,D", [(8, 8), (7, 8), (8, 3)])
def test_deep_register_tile_kernel_on_a_large_batch(depth, D, monkeypatch):
    """max_depth 7 and 8 at the shapes the variant exists for: 2^17 rows x 128 features; 10 trees (resident values: 8 KiB per tree at 8
    outputs), 19 (the most that stay resident), 45 (grouped: groups of 8 trees, the last one partial; an odd count takes the single-tree
    tail) -- default dispatch, bitwise equal to the general kernel and to the second-generation kernel."""
    n_trees = 45
    case = dict(name="prl8", seed=4300 + depth + D, N=3000, F=128, Fc=0, D=D, depth=depth, n_bins=64, score="L2", gen="Quantile", policy="oblivious", trees=n_trees)
    m, _ = _grow(case)
    assert int(np.asarray(m.get_ensemble_data()["depths"]).max()) == depth
    Xp = _batch(case, (1 << 17) + 77, seed=6)
    ranges = ((0, 10), (1, 20), (0, 19), (0, 0), (3, 45), (0, 44), (40, 45))
    outs = {}
    for mode, env in (("reg", {"GBRL_HIP_PREDICT_REG_ONLY": "1"}), ("reg_groups_of_3", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "3"}),
                      ("gen2", {"GBRL_HIP_PREDICT_NO_REG": "1", "GBRL_HIP_PREDICT_NO_PC": "1"}), ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"})):
        _set(monkeypatch, env)
        outs[mode] = [np.asarray(m.predict(Xp, None, a, b)) for a, b in ranges]
    for mode in ("reg", "reg_groups_of_3", "gen2"):
        for r, a, b in zip(ranges, outs[mode], outs["generic"]):
            assert np.array_equal(a, b), (mode, r)
    assert np.abs(outs["reg"][0]).max() > 0


def test_register_tile_kernel_takes_device_inputs_and_declines_what_it_does_not_cover(monkeypatch):
    """Device-resident inputs (the bench's path) through DLPack; a feature count that is not a multiple of four goes to the
    packed-code kernel; seven levels go to the round-5 deep variant; trees deeper than eight levels are declined by both (REG_ONLY raises)
    and predicted by the older kernels."""
    import torch
    import gbrl_amd
    case = dict(name="prd", seed=77, N=2000, F=24, Fc=0, D=4, depth=4, n_bins=32, score="L2", gen="Uniform", policy="oblivious", trees=9)
    m, _ = _grow(case)
    Xp = _batch(case, 40000, seed=1)
    _set(monkeypatch, {"GBRL_HIP_PREDICT_GENERIC": "1"})
    want = np.asarray(m.predict(Xp, None, 0, 0))
    _set(monkeypatch, {})
    t = torch.from_numpy(Xp).to("cuda:0")
    md = gbrl_amd.GBRL(**K.ctor_kwargs(case, device="cuda"))
    X, Xc, G, y = K.make_inputs(case)
    K.drive(md, case, X, Xc, G, y)
    _set(monkeypatch, {"GBRL_HIP_PREDICT_REG_ONLY": "1"})
    got = torch.from_dlpack(md.predict((t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda"), None, 0, 0)).cpu().numpy()
    assert np.array_equal(got, want)
    for variant, taken in ((dict(case, F=23, name="prd_odd"), True), (dict(case, depth=7, name="prd_deep7"), True), (dict(case, depth=9, name="prd_deep9"), False)):
        _set(monkeypatch, {})
        mo, _ = _grow(variant)
        Xo = _batch(variant, 40000, seed=2)
        _set(monkeypatch, {"GBRL_HIP_PREDICT_GENERIC": "1"})
        ref = np.asarray(mo.predict(Xo, None, 0, 0))
        _set(monkeypatch, {"GBRL_HIP_PREDICT_REG_ONLY": "1"})
        if taken:
            assert np.array_equal(np.asarray(mo.predict(Xo, None, 0, 0)), ref)
        else:
            with pytest.raises(RuntimeError, match="REG_ONLY"):
                mo.predict(Xo, None, 0, 0)
            _set(monkeypatch, {})
            assert np.array_equal(np.asarray(mo.predict(Xo, None, 0, 0)), ref)


def _cat_batch(case, n, seed):
    """rows of the case's numeric and categorical widths; one cell in eleven holds a category no tree has seen"""
    rng = np.random.default_rng(seed)
    X, Xc, _, _ = K.make_inputs(case)
    idx = rng.integers(0, (X if X is not None else Xc).shape[0], size=n)
    Xn = None if X is None else _specials(rng, np.ascontiguousarray(X[idx] + rng.standard_normal((n, X.shape[1])).astype(np.float32) * np.float32(0.05)))
    Cn = None
    if Xc is not None:
        Cn = np.ascontiguousarray(Xc[idx])
        unseen = rng.integers(0, 11, size=Cn.shape) == 0
        Cn[unseen] = b"never-seen"
    return Xn, Cn


PC_MODES = (("packed", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1"}),
            ("packed_grouped", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "1"}),
            ("packed_groups_of_3", {"GBRL_HIP_PREDICT_REG_ONLY": "1", "GBRL_HIP_PREDICT_REG_MIN_ROWS": "1", "GBRL_HIP_PREDICT_REG_GROUPED": "3"}),
            ("gen2", {"GBRL_HIP_PREDICT_NO_REG": "1", "GBRL_HIP_PREDICT_NO_PC": "1"}),
            ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"}))


@pytest.mark.parametrize("depth,D", [(3, 3), (4, 8), (6, 8), (5, 4), (6, 1)])
@pytest.mark.parametrize("F,Fc", [(9, 2), (5, 3), (0, 3), (130, 1), (200, 0), (31, 7)])
def test_packed_code_kernel_equals_the_general_kernel(F, Fc, depth, D, monkeypatch):
    """Rows as packed codes (k_pack_codes + k_predict_pc): numeric features as 16-bit rank fields (two per register, both halves
    exercised), categorical columns as inverted one-hot bits (equality conditions, cells no tree has seen, several columns per word),
    models with only categorical columns, rows wider than the fp32 bank, mixed trees -- bitwise the general kernel, over sub-ranges,
    ragged batches and both launch shapes."""
    n_trees = 21
    case = dict(name="pc", seed=1300 + 3 * depth + F + Fc, N=1500, F=F, Fc=Fc, D=D, depth=depth, n_bins=32, score="L2", gen="Uniform", policy="oblivious",
                trees=n_trees, loop="rmse" if F > 0 and D == 1 else None, y_cat_weight=0.5, n_tokens=6)
    if case["loop"] is None:
        del case["loop"]
    m, _ = _grow(case)
    e = m.get_ensemble_data()
    if Fc > 0:
        assert (np.asarray(e["is_numerics"]) == 0).any() or F > 0   # (a purely categorical model has only categorical conditions)
    ranges = ((0, 0), (0, 1), (2, 17), (5, 6), (20, 21))
    for n in (1, 65, 1000, 4133):
        Xn, Cn = _cat_batch(case, n, seed=n + F)
        outs = {}
        for mode, env in PC_MODES:
            _set(monkeypatch, env)
            outs[mode] = [np.asarray(m.predict(Xn, Cn, a, b)) for a, b in ranges]
        for mode, _ in PC_MODES:
            for r, a, b in zip(ranges, outs[mode], outs["generic"]):
                assert a.shape == b.shape and np.array_equal(a, b), (mode, n, r)
        assert np.abs(outs["packed"][0]).max() > 0


def test_packed_code_kernel_on_the_cfg5_miniature_and_what_it_declines(monkeypatch):
    """BASELINE configs[4] in miniature (24 numeric + 8 categorical columns, 320 trees grown by the rmse loop: the fixture's case), a
    large batch through the default dispatch; 330 numeric features need 165 words > the bank: declined, older kernels as before."""
    case = next(c for c in K.CASES if c["name"] == "obl_l2_u_cfg5mini")
    m, _ = _grow(case)
    Xn, Cn = _cat_batch(case, 40000, seed=9)
    outs = {}
    for mode, env in (("packed", {"GBRL_HIP_PREDICT_REG_ONLY": "1"}), ("gen2", {"GBRL_HIP_PREDICT_NO_PC": "1"}), ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"})):
        _set(monkeypatch, env)
        outs[mode] = [np.asarray(m.predict(Xn, Cn, a, b)) for a, b in ((0, 0), (17, 300), (0, 33))]
    for mode in ("packed", "gen2"):
        for a, b in zip(outs[mode], outs["generic"]):
            assert np.array_equal(a, b), mode
    wide = dict(name="pcw", seed=5, N=1200, F=330, Fc=0, D=2, depth=4, n_bins=16, score="L2", gen="Uniform", policy="oblivious", trees=5)
    _set(monkeypatch, {})
    mw, _ = _grow(wide)
    Xw = _batch(wide, 33000, seed=3)
    _set(monkeypatch, {"GBRL_HIP_PREDICT_GENERIC": "1"})
    ref = np.asarray(mw.predict(Xw, None, 0, 0))
    _set(monkeypatch, {"GBRL_HIP_PREDICT_REG_ONLY": "1"})
    with pytest.raises(RuntimeError, match="REG_ONLY"):
        mw.predict(Xw, None, 0, 0)
    _set(monkeypatch, {})
    assert np.array_equal(np.asarray(mw.predict(Xw, None, 0, 0)), ref)


def test_random_sweep_of_the_register_tile_kernels_has_no_mismatch():
    """120 random ensembles / batches / tree ranges (scripts/predict_reg_sweep.py: 1-70 trees, depth 1-6, 1-16 outputs, up to 300 numeric
    and 9 categorical columns, special float values, unseen categories): register-tile kernels == general kernel, byte for byte."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "predict_reg_sweep.py"), "120", "424242"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 mismatches" in out.stdout
