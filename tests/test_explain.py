"""Inspection methods: tree_shap / ensemble_shap / export / print_tree /
print_ensemble_metadata / plot_tree of the drop-in class against fixtures made by the REFERENCE's CPU build
(tests/golden/make_explain_golden.py).  The model is LOADED from the reference's own .gbrl_model bytes, so these run
without a GPU: the methods are served from the host copy of the ensemble, like the reference's (gbrl.cpp:1269-1391).

Bars: exported header and printed text byte-identical; SHAP within 1e-5 of the largest |value| of the array (the
recursion is float32 with cancellation, so the error is measured against the array's scale, not element-wise)."""
import os

import numpy as np
import pytest

import cases as K
import gbrl_amd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SHAP_TOL = 1e-5


def _load(name, tmp_path):
    g = np.load(os.path.join(GOLDEN, "explain_" + name + ".npz"))
    case = K.BY_NAME[name]
    X, Xc, G, y = K.make_inputs(case)
    assert K.inputs_digest(X, Xc, G, y) == str(g["inputs_sha256"]), "input synthesis drifted from the fixture"
    p = tmp_path / "ref.gbrl_model"
    p.write_bytes(g["model_file"].tobytes())
    m = gbrl_amd.GBRL.load(str(p))
    n = K.EXPLAIN_ROWS
    xs = None if X is None else np.ascontiguousarray(X[:n])
    xcs = None if Xc is None else np.ascontiguousarray(Xc[:n])
    return case, g, m, xs, xcs


def _close(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == np.float32, what
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) / scale
    assert err <= SHAP_TOL, f"{what}: {err:.3g} of the array's scale"


@pytest.mark.parametrize("name", list(K.EXPLAIN_CASES))
def test_shap_matches_the_reference(name, tmp_path):
    case, g, m, xs, xcs = _load(name, tmp_path)
    base, norm, offset = g["base_poly"], g["norm_values"], g["offset"]
    b2, n2, o2 = K.poly_vectors(case["depth"])          # the restated Python-layer helper reproduces the stored vectors
    assert np.array_equal(b2, base) and np.array_equal(n2, norm) and np.array_equal(o2, offset)
    T = int(g["n_trees"])
    for t in sorted({0, T // 2, T - 1}):
        _close(m.tree_shap(t, xs, xcs, norm, base, offset), g["shap_tree_%d" % t], f"tree {t}")
    _close(m.ensemble_shap(xs, xcs, norm, base, offset), g["shap_ensemble"], "ensemble")
    one = m.tree_shap(0, None if xs is None else xs[0], None if xcs is None else xcs[0], norm, base, offset)
    assert one.shape == (1, case["F"] + case.get("Fc", 0), case["D"])     # a 1-D input is ONE sample (binding.cpp:996-1003)
    _close(one, g["shap_one_row"], "single row")
    # the ensemble's SHAP values are the per-tree values summed in tree order
    acc = np.zeros_like(g["shap_ensemble"])
    for t in range(T):
        acc += m.tree_shap(t, xs, xcs, norm, base, offset)
    _close(acc, g["shap_ensemble"], "sum over trees")


@pytest.mark.parametrize("name", [n for n, v in K.EXPLAIN_CASES.items() if v])
def test_exported_header_is_byte_identical(name, tmp_path):
    _, g, m, _, _ = _load(name, tmp_path)
    assert int(m.get_ensemble_data()["alloc_data_size"]) == int(g["alloc_data_size"])
    for k, (mname, fmt, typ, prefix) in enumerate(K.EXPLAIN_CASES[name]):
        h = tmp_path / ("m%d.h" % k)
        assert m.export(str(h), mname, fmt, typ, prefix) == 0
        assert h.read_bytes() == g["export_%d" % k].tobytes(), (name, fmt, typ)


def test_export_error_behaviour(tmp_path):
    _, _, grd, _, _ = _load("grd_cos_q_ac", tmp_path)
    with pytest.raises(RuntimeError, match="Export is supported only for Oblivious trees"):
        grd.export(str(tmp_path / "g.h"))
    _, _, obl, _, _ = _load("obl_l2_q", tmp_path)
    with pytest.raises(RuntimeError, match="Invalid exportFormat"):
        obl.export(str(tmp_path / "a.h"), "", "fp64", "full", "")
    with pytest.raises(RuntimeError, match="Invalid exportType"):
        obl.export(str(tmp_path / "a.h"), "", "float", "tiny", "")
    with pytest.raises(RuntimeError, match="File opening error"):
        obl.export(str(tmp_path / "no_such_dir" / "a.h"))


@pytest.mark.parametrize("name", list(K.EXPLAIN_CASES))
def test_printed_text_is_byte_identical(name, tmp_path, capfdbinary):
    _, g, m, _, _ = _load(name, tmp_path)
    capfdbinary.readouterr()
    m.print_ensemble_metadata()
    assert capfdbinary.readouterr().out == g["print_meta"].tobytes()
    m.print_tree(0)
    assert capfdbinary.readouterr().out == g["print_tree_0"].tobytes()
    m.print_tree()
    assert capfdbinary.readouterr().out == g["print_tree_last"].tobytes()
    with pytest.raises(RuntimeError, match="Invalid tree index"):
        m.print_tree(int(g["n_trees"]) + 3)


def test_shap_argument_checks_and_plot(tmp_path):
    case, g, m, xs, xcs = _load("obl_l2_q_cat", tmp_path)
    base, norm, offset = g["base_poly"], g["norm_values"], g["offset"]
    with pytest.raises(RuntimeError, match="Invalid tree index"):
        m.tree_shap(99, xs, xcs, norm, base, offset)
    with pytest.raises(RuntimeError, match="Incompatible dimensions"):
        m.tree_shap(0, xs[:, :3].copy(), xcs, norm, base, offset)
    with pytest.raises(RuntimeError, match="C-contiguous"):
        m.tree_shap(0, xs[:, ::-1], xcs, norm, base, offset)
    with pytest.raises(RuntimeError, match="max_depth"):
        m.tree_shap(0, xs, xcs, norm[:2], base, offset)
    with pytest.raises(RuntimeError, match="without Graphviz"):       # gbrl.cpp:1541-1546 in a build without Graphviz
        m.plot_tree(0, str(tmp_path / "t"))
    empty = gbrl_amd.GBRL(input_dim=3, output_dim=2, policy_dim=2, max_depth=3, grow_policy="oblivious", device="cpu")
    b, n, o = K.poly_vectors(3)
    z = empty.ensemble_shap(np.zeros((4, 3), np.float32), None, n, b, o)    # no trees: zeros (gbrl.cpp:1306-1330)
    assert z.shape == (4, 3, 2) and not z.any()
