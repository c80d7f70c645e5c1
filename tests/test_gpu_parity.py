"""GPU parity tests (run on the MI355X box with -m gpu): the HIP product, called through the gbrl_cpp binding which
goes through the C ABI only, against (a) the committed golden vectors produced by the reference's CPU path and (b) the
oracle restatement on fresh seeds.

Bar (BASELINE.json north_star): tree STRUCTURE bit-identical (split feature indices, threshold bits, directions, leaf
layout); leaf values and predictions within 1e-5 relative, where "relative" is |a-b| / max(|b|, gradient scale)
(SURVEY.md hard part 5: the oracle's own float32 sequential sums limit what a relative error of a near-zero mean means).
"""
import os

import numpy as np
import pytest

import cases as K
from helpers import assert_structure_equal, assert_values_close, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _torch_input(keep):
    import torch

    def to_input(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        keep.append(t)
        return (t.data_ptr(), tuple(t.shape), str(t.dtype), "cuda")
    return to_input


def _from_capsule(c):
    import torch
    if isinstance(c, np.ndarray):
        return c
    return torch.from_dlpack(c).cpu().numpy()


def _run_product(case, X, Xc, G, y, device):
    import gbrl_amd
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case, device=device))
    if device == "cpu":      # host buffers in, NumPy out (compute is on the GPU regardless)
        pred = K.drive(m, case, X, Xc, G, y)
    else:                    # torch device tensors in as 4-tuples, DLPack (kDLROCM) out
        keep = []
        pred = K.drive(m, case, X, Xc, G, y, to_input=_torch_input(keep), to_numpy=_from_capsule)
    return m, np.asarray(pred)


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES if not c.get("fragile") and not c.get("long_loop") and not c.get("neartie")])
def test_product_matches_reference_golden(name):
    case, g, (X, Xc, G, y) = load_golden(name)
    m, pred = _run_product(case, X, Xc, G, y, "cpu")
    e = m.get_ensemble_data()
    assert m.get_num_trees() == int(g["n_trees"]) and m.get_iteration() == int(g["iteration"])
    assert_structure_equal(e, g, what=name + ": ")
    scale = float(np.abs(G).mean()) if y is None else float(np.abs(y).mean())
    assert_values_close(e, g, scale, TOL, what=name + ": ")
    assert rel_err(pred.reshape(g["pred"].shape), g["pred"], scale) <= TOL


@pytest.mark.parametrize("name", [c["name"] for c in K.FIT_CASES])
def test_fit_matches_reference_golden(name):
    """GBRL.fit (gbrl.cpp:983-1104, fitter.cpp:117-261): candidates from the whole data set, one tree per batch, MultiRMSE.
    Structure bit-identical to the reference's fit(); bias / leaf values / predictions / returned loss within 1e-5 (the
    reference's bias is a thread-count dependent float32 mean)."""
    import gbrl_amd
    case, g, (X, Xc, G, y) = load_golden(name)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    loss, pred = K.drive_fit(m, case, X, y, Xc)
    e = m.get_ensemble_data()
    assert m.get_num_trees() == int(g["n_trees"]) == case["fit_iterations"] and m.get_iteration() == int(g["iteration"])
    assert_structure_equal(e, g)
    scale = float(np.abs(y).mean())
    assert_values_close(e, g, scale, TOL)
    assert rel_err(np.asarray(m.get_bias()), g["bias"], scale) <= TOL
    assert rel_err(pred, g["pred"], scale) <= TOL
    assert abs(loss - float(g["fit_loss"])) <= TOL * max(1.0, abs(float(g["fit_loss"])))
    # the returned loss is the MultiRMSE of the final predictions
    want = np.sqrt(0.5 * float(((pred.reshape(len(X), -1) - y.reshape(len(X), -1)).astype(np.float64) ** 2).sum()) / len(X))
    assert abs(loss - want) <= 1e-5 * max(1.0, want)


def test_fit_api_behaviour():
    import gbrl_amd
    case, g, (X, Xc, G, y) = load_golden("fit_obl_l2_q")
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    with pytest.raises(RuntimeError, match="Invalid loss function"):
        m.fit(X, None, y, 2, False, "L1")
    with pytest.raises(RuntimeError, match="Targets output dim"):
        m.fit(X, None, y[:, :1], 2, False, "MultiRMSE")
    with pytest.raises(RuntimeError, match="Number of observations"):
        m.fit(X[:-1], None, y, 2, False, "MultiRMSE")
    l1, _ = K.drive_fit(m, case, X, y)
    # shuffled fit: same data in another order -> a model of comparable quality, one tree per iteration, bias = mean(targets)
    m2 = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    m2.set_feature_weights(np.ones(case["F"], np.float32))
    for o in K.optimizers(case):
        m2.set_optimizer(**o)
    m2.set_feature_mapping(np.arange(case["F"], dtype=np.int32), np.ones(case["F"], dtype=bool))
    l2 = m2.fit(X, None, y, case["fit_iterations"], True, "MultiRMSE")
    assert m2.get_num_trees() == case["fit_iterations"]
    assert np.allclose(np.asarray(m2.get_bias()), y.mean(axis=0), rtol=1e-5, atol=1e-6)
    assert abs(l2 - l1) < 0.25 * l1
    # fit() continues an existing ensemble: more iterations keep lowering the training loss
    l3 = m.fit(X, None, y, 4, False, "MultiRMSE")
    assert m.get_num_trees() == case["fit_iterations"] + 4 and np.isfinite(l3)


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES if c.get("fragile")])
def test_fragile_case_is_exact_or_an_explained_near_tie(name):
    """Inputs on which the reference disagrees with ITSELF between OMP_NUM_THREADS 3 and 8 (make_golden.py flags them):
    the product must either reproduce the 8-thread fixture exactly or differ first at a split whose two candidates
    are tied within the reference's own float32 summation noise (neartie.py), with the product holding the true max."""
    import neartie
    import oracle
    case, g, (X, Xc, G, y) = load_golden(name)
    assert not bool(g["ref_stable_across_threads"])
    m, pred = _run_product(case, X, Xc, G, y, "cpu")
    e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    mm = neartie.first_mismatch(g, e, case["policy"])
    if mm is None:
        return
    t = mm[0]
    if y is not None:   # gradients of tree t in the rmse loop: pred(t trees) - y, reproduced with the bit-exact restatement
        o = oracle.OracleGBRL(**K.ctor_kwargs(case))
        K.drive(o, dict(case, trees=t), X, Xc, G, y)
        G_t = (np.asarray(o.predict(X, Xc, 0, 0)).astype(np.float32).reshape(y.shape) - y) if t else (0 - y)
    else:
        G_t = G
    info = neartie.explain_first_mismatch(case, X, Xc, G_t, g, e)
    print("near-tie:", info)
    assert info["explained"], info
    assert info["product_is_true_max"], info
    # the ONE known difference of this fixture (rounds 4-6): tree 0, an 896-row node, gap 1.6e-7 relative -- an L2 node where the reference's
    # thread-count-dependent standardisation decides (it disagrees with itself between 3 and 8 threads there); anything else is new
    assert info["tree"] == 0 and info["n_rows"] == 896 and info["gap_rel"] < 2e-7, info


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES if c.get("neartie")])
def test_near_tie_specimen_is_exact_with_the_replay_and_explained_without(name, monkeypatch):
    """Inputs on which the reference -- stably, at every thread count -- picks a candidate whose float64 score is below the
    maximum by less than its own float32 summation noise: the product holds the true maximum; everything before that split
    must be bit-identical."""
    import neartie
    case, g, (X, Xc, G, y) = load_golden(name)
    monkeypatch.delenv("GBRL_HIP_NO_NEARTIE_REPLAY", raising=False)
    m, pred = _run_product(case, X, Xc, G, y, "cpu")
    e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    # round 6 (no escape hatch without a count): with the near-tie replay -- the default -- the specimen is EXACT ...
    assert neartie.first_mismatch(g, e, case["policy"]) is None, "the near-tie replay no longer reproduces the reference's choice on this specimen"
    assert_structure_equal(e, g)
    # ... and without it, it is exactly the known near-tie: tree 0, the product on the true maximum, the gap inside the reference's noise
    monkeypatch.setenv("GBRL_HIP_NO_NEARTIE_REPLAY", "1")
    m, pred = _run_product(case, X, Xc, G, y, "cpu")
    e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    info = neartie.explain_first_mismatch(case, X, Xc, G, g, e)
    print("near-tie without the replay:", info)
    assert info is not None and info["explained"] and info["product_is_true_max"] and info["tree"] == 0, info


def test_cfg5_miniature_grows_the_reference_ensemble():
    """BASELINE configs[4] in miniature (24 numeric + 8 categorical columns, UNIFORM candidates, oblivious depth 6, 320 trees grown
    by the rmse loop): the product must reproduce the reference's ensemble; over a loop this long the first difference, if any,
    must be an explained near-tie (the reference's float32 summation noise, neartie.py) with the product holding the true maximum,
    and every tree before it must be bit-identical."""
    import neartie
    import oracle
    case, g, (X, Xc, G, y) = load_golden("obl_l2_u_cfg5mini")
    m, pred = _run_product(case, X, Xc, G, y, "cpu")
    e = {k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    assert m.get_num_trees() == int(g["n_trees"]) == 320
    mm = neartie.first_mismatch(g, e, case["policy"])
    scale = float(np.abs(y).mean())
    # round 6: all 320 trees are bit-identical on the final builds of rounds 5 and 6; a difference -- even an explainable
    # one -- fails here, with its analysis in the message
    if mm is not None:
        t = mm[0]
        o = oracle.OracleGBRL(**K.ctor_kwargs(case))
        K.drive(o, dict(case, trees=t), X, Xc, G, y)
        G_t = np.asarray(o.predict(X, Xc, 0, 0)).astype(np.float32).reshape(y.shape) - y
        info = neartie.explain_first_mismatch(case, X, Xc, G_t, g, e)
        raise AssertionError("cfg5mini: first structural difference at tree %d of 320: %r" % (t, info))
    assert_structure_equal(e, g)
    assert_values_close(e, g, scale, TOL)
    assert rel_err(pred.reshape(g["pred"].shape), g["pred"], scale) <= TOL


@pytest.mark.parametrize("device", ["cpu", "cuda"])
@pytest.mark.parametrize("generic", [False, True])
def test_cfg5_miniature_predict_on_the_reference_model(device, generic, tmp_path, monkeypatch):
    """configs[4] in miniature, predict only: the model FILE the reference wrote (320 oblivious depth-6 trees, numeric and
    categorical conditions) is loaded by the product; predictions over the whole ensemble and over sub-ranges of it
    (start_tree_idx / stop_tree_idx) must match the reference's within 1e-5, for host and device inputs, through the fast kernel and
    through the general one -- and the two kernels must agree bit for bit."""
    import gbrl_amd
    case, g, (X, Xc, G, y) = load_golden("obl_l2_u_cfg5mini")
    path = str(tmp_path / "ref.gbrl_model")
    open(path, "wb").write(np.asarray(g["model_file"]).tobytes())
    if generic:
        monkeypatch.setenv("GBRL_HIP_PREDICT_GENERIC", "1")
    m = gbrl_amd.GBRL.load(path)
    assert m.get_num_trees() == 320
    if device == "cuda":
        m.to_device("cuda")
    keep = []
    xi = _torch_input(keep)(X) if device == "cuda" else X
    scale = float(np.abs(y).mean())
    got = {}
    for a, b in [(0, 0)] + [tuple(r) for r in case["pred_ranges"]]:
        p = _from_capsule(m.predict(xi, Xc, a, b))
        want = g["pred"] if (a, b) == (0, 0) else g["pred_%d_%d" % (a, b)]
        assert rel_err(p.reshape(want.shape), want, scale) <= TOL, (a, b)
        got[(a, b)] = p
    if generic:
        monkeypatch.delenv("GBRL_HIP_PREDICT_GENERIC")
        for (a, b), p in got.items():
            q = _from_capsule(m.predict(xi, Xc, a, b))
            if (b if b else 320) - a < 128:
                assert np.array_equal(q, p), (a, b)       # one chain per row in tree order: the same bits from both kernels
            else:
                # 768 rows x >= 128 trees: the fast path spreads tree ranges over blocks and adds the partial sums in tree order
                # (kern::predict) -- a different association of the same float32 terms
                assert rel_err(q, p, scale) <= 2e-6, (a, b)


def test_configs4_full_width_properties():
    """BASELINE configs[4] at its full WIDTH (192 numeric + 64 categorical columns = 8 KiB of S128 cells per row, uniform candidates,
    oblivious depth 6) on 2^16 rows, through size-independent properties: (1) one step with lr = 1: predict returns minus the
    value of the leaf the row is routed to by the stored conditions, numeric and categorical, bit for bit; (2) the leaf values are
    the exact means of the raw gradients; (3) a categorical condition was actually chosen (the target depends on categorical
    columns); (4) an ensemble of 40 trees grown on 4096-row minibatches predicts the same bits through the fast and the general
    kernel, over the whole range and over sub-ranges, from host cells and from device-resident inputs."""
    import gbrl_amd
    import torch
    rng = np.random.default_rng(5)
    N, F, Fc, D = 1 << 16, 192, 64, 8
    X = rng.standard_normal((N, F), dtype=np.float32)
    Xc = K.TOKENS[rng.integers(0, 32, size=(N, Fc))]
    W = rng.standard_normal((8, D)).astype(np.float32)
    G = (np.tanh(X[:, :8] @ W) + 0.5 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
    Xc[:, 5] = K.TOKENS[rng.integers(0, 2, size=N)]      # a two-valued categorical column that carries a strong signal
    G += ((Xc[:, 5] == K.TOKENS[1]).astype(np.float32) * np.float32(3.0))[:, None]
    case = dict(name="c5", seed=0, N=N, F=F, Fc=Fc, D=D, depth=6, n_bins=256, score="L2", gen="Uniform", policy="oblivious", trees=1,
                opts=[dict(algo="SGD", scheduler="Const", init_lr=1.0, start_idx=0, stop_idx=D)])
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    pred = np.asarray(K.drive(m, case, X, Xc, G, None))
    e = m.get_ensemble_data()
    depth = int(e["depths"][0])
    assert depth == 6
    fi, fv, isn = np.asarray(e["feature_indices"])[0], np.asarray(e["feature_values"])[0], np.asarray(e["is_numerics"])[0]
    cv = np.asarray(e["categorical_values"])[0]
    leaf = np.zeros(N, np.int64)
    for d in range(depth):
        right = (X[:, fi[d]] > fv[d]) if isn[d] else (Xc[:, fi[d]] == cv[d])
        leaf |= right.astype(np.int64) << (depth - 1 - d)
    vals = np.asarray(e["values"])
    assert np.array_equal(pred, (np.float32(0) - vals[leaf]).astype(np.float32))           # (1)
    for l in range(64):                                                                     # (2)
        rows = leaf == l
        if rows.any():
            want = G[rows].astype(np.float64).mean(axis=0)
            assert np.max(np.abs(vals[l] - want) / np.maximum(np.abs(want), 0.5)) < 1e-6
    assert not isn[:depth].all(), "no categorical condition was chosen"                   # (3)
    # (4) a small ensemble on minibatches, predicted four ways
    for i in range(39):
        o = (i * 4096) % (N - 4096)
        gi = (G[o:o + 4096] * np.float32(1.0 / (1 + i))).astype(np.float32)
        m.step(np.ascontiguousarray(X[o:o + 4096]), np.ascontiguousarray(Xc[o:o + 4096]), gi)
    assert m.get_num_trees() == 40
    xt = torch.from_numpy(X).to("cuda:0")
    xin = (xt.data_ptr(), tuple(xt.shape), str(xt.dtype), "cuda")
    outs = {}
    for generic in ("0", "1"):
        os.environ["GBRL_HIP_PREDICT_GENERIC"] = generic
        try:
            for rng_ in [(0, 0), (0, 1), (3, 17), (39, 40)]:
                outs[(generic, "host") + rng_] = np.asarray(m.predict(X, Xc, *rng_))
                outs[(generic, "dev") + rng_] = np.asarray(m.predict(xin, Xc, *rng_))
        finally:
            os.environ.pop("GBRL_HIP_PREDICT_GENERIC", None)
    for rng_ in [(0, 0), (0, 1), (3, 17), (39, 40)]:
        ref = outs[("1", "host") + rng_]
        for k in (("0", "host"), ("0", "dev"), ("1", "dev")):
            assert np.array_equal(outs[k + rng_], ref), (k, rng_)


@pytest.mark.parametrize("name", ["obl_l2_q", "grd_cos_q_ac", "cfg1_rmse_loop"])
def test_torch_device_tuple_path_equals_host_path(name):
    case, g, (X, Xc, G, y) = load_golden(name)
    m1, p1 = _run_product(case, X, Xc, G, y, "cpu")
    m2, p2 = _run_product(case, X, Xc, G, y, "cuda")
    e1, e2 = m1.get_ensemble_data(), m2.get_ensemble_data()
    for k in K.ENSEMBLE_KEYS:
        assert np.array_equal(np.asarray(e1[k]), np.asarray(e2[k])), k   # same bytes, whichever way the data came in
    assert np.array_equal(p1.reshape(-1), p2.reshape(-1))
    assert m2.get_device() == "cuda"


def test_same_inputs_same_bytes():
    """Determinism: integer accumulation makes every run produce identical bytes (the reference's CUDA path does not)."""
    case, g, (X, Xc, G, y) = load_golden("obl_l2_q_d6")
    runs = [_run_product(case, X, Xc, G, y, "cpu") for _ in range(3)]
    for m, p in runs[1:]:
        for k in K.ENSEMBLE_KEYS:
            assert np.array_equal(np.asarray(m.get_ensemble_data()[k]), np.asarray(runs[0][0].get_ensemble_data()[k])), k
        assert np.array_equal(p, runs[0][1])


@pytest.mark.parametrize("seed", [201, 202, 203, 204])
@pytest.mark.parametrize("policy,score,gen", [("oblivious", "L2", "Quantile"), ("greedy", "Cosine", "Quantile"),
                                              ("greedy", "L2", "Uniform"), ("oblivious", "Cosine", "Uniform")])
def test_product_matches_oracle_on_fresh_seeds(seed, policy, score, gen):
    import oracle
    case = dict(name="fresh", seed=seed, N=3000, F=10, Fc=0, D=4, depth=5, n_bins=128, score=score, gen=gen,
                policy=policy, trees=3)
    X, Xc, G, y = K.make_inputs(case)
    m, pred = _run_product(case, X, Xc, G, y, "cpu")
    ref = oracle.OracleGBRL(**K.ctor_kwargs(case))
    pref = K.drive(ref, case, X, Xc, G, y)
    e, r = m.get_ensemble_data(), ref.get_ensemble_data()
    assert_structure_equal(e, r)
    scale = float(np.abs(G).mean())
    assert_values_close(e, r, scale, TOL)
    assert rel_err(pred, pref, scale) <= TOL


def test_quantile_thresholds_bit_exact_at_scale():
    """A3 at a size where the oracle's sort is still cheap: thresholds chosen by the GPU's exact selection are data
    values at the reference's ranks -- checked through the tree (every stored threshold must be one of them) and the
    early-stop/ragged path (N not a multiple of n_bins+1, heavy duplicates, a constant column)."""
    import gbrl_amd
    rng = np.random.default_rng(7)
    N, F, B = 50021, 12, 256
    X = rng.standard_normal((N, F)).astype(np.float32)
    X[:, 3] = np.round(X[:, 3])          # ~7 distinct values
    X[:, 5] = 1.5                        # constant
    X[::7, 6] = -0.0                     # signed zeros
    G = (np.sign(X[:, :2]) + 0.1 * rng.standard_normal((N, 2))).astype(np.float32)
    case = dict(name="q", seed=0, N=N, F=F, D=2, depth=6, n_bins=B, score="L2", gen="Quantile", policy="greedy", trees=1)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, None, G, None)
    e = m.get_ensemble_data()
    counts = np.full(B + 1, N // (B + 1)); counts[: N % (B + 1)] += 1
    ranks = np.cumsum(counts)[:B] - 1
    allowed = {f: set(np.sort(X[:, f])[ranks].view(np.uint32).tolist()) | {0, 0x80000000} for f in range(F)}
    fi, fv, dep = np.asarray(e["feature_indices"]), np.asarray(e["feature_values"]), np.asarray(e["depths"])
    for leaf in range(fi.shape[0]):
        for d in range(dep[leaf]):
            assert int(fv[leaf, d].view(np.uint32)) in allowed[int(fi[leaf, d])]


def test_full_size_properties_config2_shape():
    """BASELINE configs[1] at its FULL size (oblivious / L2 / quantile, N=2^20, F=128, D=8, depth 6) through size-independent
    properties: (1) predict(X) after one tree with lr=1 returns -(leaf mean) so the per-leaf means of G reproduce the
    stored values; (2) leaf counts from edge weights multiply to the leaf population; (3) the tree is a proper
    oblivious tree (same condition per level for all leaves); (4) adding a constant to G leaves the structure unchanged
    (L2 standardisation is shift invariant)."""
    import gbrl_amd
    rng = np.random.default_rng(0)
    N, F, D = 1 << 20, 128, 8
    X = rng.standard_normal((N, F), dtype=np.float32)
    W = rng.standard_normal((8, D)).astype(np.float32)
    G = (np.tanh(X[:, :8] @ W) + 0.5 * rng.standard_normal((N, D))).astype(np.float32)
    case = dict(name="c2", seed=0, N=N, F=F, D=D, depth=6, n_bins=256, score="L2", gen="Quantile", policy="oblivious",
                trees=1, opts=[dict(algo="SGD", scheduler="Const", init_lr=1.0, start_idx=0, stop_idx=D)])
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    pred = np.asarray(K.drive(m, case, X, None, G, None))
    e = m.get_ensemble_data()
    depth = int(e["depths"][0])
    assert depth == 6 and e["values"].shape == (64, D)
    fi, fv = np.asarray(e["feature_indices"])[0], np.asarray(e["feature_values"])[0]
    leaf = np.zeros(N, np.int64)
    for d in range(depth):
        leaf |= (X[:, fi[d]] > fv[d]).astype(np.int64) << (depth - 1 - d)
    vals = np.asarray(e["values"])
    assert np.array_equal(pred, (np.float32(0) - vals[leaf]).astype(np.float32))          # (1a) routing
    for l in range(64):                                                                    # (1b) exact leaf means
        rows = leaf == l
        if rows.any():
            want = G[rows].astype(np.float64).mean(axis=0)
            assert np.max(np.abs(vals[l] - want) / np.maximum(np.abs(want), 0.5)) < 1e-6
    ew = np.asarray(e["edge_weights"])                                                    # (2)
    assert np.allclose(np.prod(ew.astype(np.float64), axis=1) * N, np.bincount(leaf, minlength=64), rtol=1e-5, atol=0.5)
    ineq = np.asarray(e["inequality_directions"])                                         # (3)
    for l in range(64):
        assert [int(b) for b in ineq[l]] == [(l >> (depth - 1 - d)) & 1 for d in range(depth)]
    case2 = dict(case, name="c2s")
    m2 = gbrl_amd.GBRL(**K.ctor_kwargs(case2))                                            # (4)
    K.drive(m2, case2, X, None, (G + np.float32(3.0)).astype(np.float32), None)
    e2 = m2.get_ensemble_data()
    assert np.array_equal(np.asarray(e2["feature_indices"]), np.asarray(e["feature_indices"]))
    assert np.array_equal(np.asarray(e2["feature_values"]), np.asarray(e["feature_values"]))


def test_full_size_properties_config3_shape():
    """BASELINE configs[2] at its FULL size (greedy / Cosine / quantile, N=2^20, F=128, D=8 = policy [0,7) lr 0.1 + value [7,8)
    lr 0.01, depth 6): (1) the leaves partition the rows (every row satisfies exactly one leaf's path); (2) leaf values are the
    exact means of the raw gradients over the leaf's rows; (3) edge weights multiply to the leaf's share of the rows; (4) predict
    returns bias - lr_k * value[leaf(row)] per optimiser range, bit for bit."""
    import gbrl_amd
    rng = np.random.default_rng(1)
    N, F, D = 1 << 20, 128, 8
    X = rng.standard_normal((N, F), dtype=np.float32)
    W = rng.standard_normal((8, D)).astype(np.float32)
    G = (np.tanh(X[:, :8] @ W) + 0.5 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
    case = dict(name="c3", seed=0, N=N, F=F, D=D, depth=6, n_bins=256, score="Cosine", gen="Quantile", policy="greedy", trees=1,
                opts=[dict(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=7),
                      dict(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=7, stop_idx=8)])
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    pred = np.asarray(K.drive(m, case, X, None, G, None))
    e = m.get_ensemble_data()
    fi, fv = np.asarray(e["feature_indices"]), np.asarray(e["feature_values"])
    dirs, dep, vals = np.asarray(e["inequality_directions"]), np.asarray(e["depths"]), np.asarray(e["values"])
    L = vals.shape[0]
    assert 2 <= L <= 64 and int(dep.max()) <= 6
    owner = np.full(N, -1, np.int64)
    hits = np.zeros(N, np.int32)
    cols = {}
    for leaf in range(L):
        ok = np.ones(N, bool)
        for d in range(int(dep[leaf])):
            key = (int(fi[leaf, d]), float(fv[leaf, d]))
            if key not in cols:
                cols[key] = X[:, key[0]] > np.float32(key[1])
            ok &= cols[key] == bool(dirs[leaf, d])
        hits += ok
        owner[ok] = leaf
    assert hits.min() == 1 and hits.max() == 1                                             # (1)
    cnt = np.bincount(owner, minlength=L)
    for leaf in range(L):                                                                  # (2)
        if cnt[leaf]:
            want = G[owner == leaf].astype(np.float64).mean(axis=0)
            assert np.max(np.abs(vals[leaf] - want) / np.maximum(np.abs(want), 0.5)) < 1e-6
    ew = np.asarray(e["edge_weights"]).astype(np.float64)                                  # (3)
    share = np.array([np.prod(ew[leaf, :int(dep[leaf])]) for leaf in range(L)])
    assert np.allclose(share * N, cnt, rtol=1e-5, atol=0.5)
    lr = np.array([0.1] * 7 + [0.01], np.float32)                                          # (4)
    want_pred = (np.float32(0) - lr[None, :] * vals[owner]).astype(np.float32)
    assert np.array_equal(pred, want_pred)


@pytest.mark.parametrize("N", [70001, 70000, 3001, 262148])   # 70000: n % 4 == 0 and n >= 65536 -> the transpose counts the first radix digit (k_transpose_count); 262148: a partial last strip inside one wave of the radix passes
def test_fast_quantile_path_equals_bisection_path(N, monkeypatch):
    """The exact selections -- LDS sort of the whole column (small batches), MSD radix multi-select (radix_select.hip),
    sample-splitter selection (quantile.hip) and 32-pass bisection (kernels.hip) -- give identical trees, also on columns built
    to stress each of them."""
    import gbrl_amd
    rng = np.random.default_rng(11)
    F = 20
    X = rng.standard_normal((N, F)).astype(np.float32)
    X[:, 1] = np.round(X[:, 1] * 3) / 3          # heavy duplicates -> equality classes / one radix slot to the last digit
    X[:, 2] = (rng.random(N) < 0.97) * 1.0       # one value holds 97 % of the column
    X[:, 4] = np.exp(3 * X[:, 4])                # heavy tail
    X[:, 7] = 0.0
    X[:, 8] = np.float32(1.0) + np.arange(N, dtype=np.float32) * np.float32(2.0 ** -23)   # all keys share the top 12+ bits
    X[:, 9] = (rng.integers(0, 2, N) * 2 - 1) * np.float32(1e-38) * rng.random(N).astype(np.float32)   # denormals, both signs
    X[:, 10] = np.array([-np.inf, -3e38, -0.0, 0.0, 3e38, np.inf], np.float32)[rng.integers(0, 6, N)]   # extremes, few values
    G = (np.tanh(X[:, :3]) + 0.3 * rng.standard_normal((N, 3))).astype(np.float32)
    case = dict(name="qq", seed=0, N=N, F=F, D=3, depth=6, n_bins=256, score="Cosine", gen="Quantile", policy="greedy", trees=2)
    outs = []
    # default (LDS sort for N <= 16384, radix above -- with the first digit counted inside the transpose when n % 4 == 0 and
    # n >= 65536), radix forced, sample splitters, radix with the separate first counting pass, bisection (last: the reference point)
    hooks = ("GBRL_HIP_QUANTILE_RADIX", "GBRL_HIP_QUANTILE_SAMPLE", "GBRL_HIP_FORCE_BISECTION", "GBRL_HIP_TRANSPOSE_COUNT")
    for env in ({}, {"GBRL_HIP_QUANTILE_RADIX": "1"}, {"GBRL_HIP_QUANTILE_SAMPLE": "1"}, {"GBRL_HIP_TRANSPOSE_COUNT": "0"}, {"GBRL_HIP_FORCE_BISECTION": "1"}):
        for k in hooks:
            monkeypatch.setenv(k, env.get(k, "1" if k == "GBRL_HIP_TRANSPOSE_COUNT" else "0"))
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        K.drive(m, case, X, None, G, None)
        outs.append({k: np.asarray(v) for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS})
    for k in K.ENSEMBLE_KEYS:
        for i, what in enumerate(("default", "radix", "sample", "separate first pass")):
            assert np.array_equal(outs[i][k], outs[4][k]), (what + " vs bisection", k)


def test_large_batch_selection_paths_agree_on_random_shapes():
    """scripts/selfcheck_sweep.py in the suite: at sizes the brute-force oracle cannot reach (65 536 ... 262 148 rows, discrete / constant /
    heavy-tailed columns) the default path, the separate first counting pass and the 32-pass bisection must grow identical trees and every
    threshold must be a rank-exact data value.  Round 3: this sweep found a round-2 bug of the radix passes (a partial last strip inside
    one wave lost queued keys at N = 2^18 + 4); seeds 7000.. contain the two cases that exposed it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "selfcheck_sweep.py"), "300", "7000"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "300 cases, 0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_more_rows_than_one_round_of_histogram_chunks():
    """N > 32 x 65536 rows: a level needs several rounds of histogram chunks (the per-block row cap is fixed by the
    fixed-point scale).  Leaf populations must add up and leaf values must be the exact leaf means."""
    import gbrl_amd
    rng = np.random.default_rng(5)
    N, F, D = 3 * (1 << 20) + 17, 16, 2
    X = rng.standard_normal((N, F), dtype=np.float32)
    G = (np.sign(X[:, :D]) + 0.25 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
    case = dict(name="big", seed=0, N=N, F=F, D=D, depth=4, n_bins=256, score="L2", gen="Quantile", policy="oblivious", trees=1,
                opts=[dict(algo="SGD", scheduler="Const", init_lr=1.0, start_idx=0, stop_idx=D)])
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    pred = np.asarray(K.drive(m, case, X, None, G, None))
    e = m.get_ensemble_data()
    depth = int(e["depths"][0])
    assert depth == 4
    fi, fv = np.asarray(e["feature_indices"])[0], np.asarray(e["feature_values"])[0]
    assert set(fi[:2].tolist()) <= {0, 1}          # the signal lives in the first two features
    leaf = np.zeros(N, np.int64)
    for d in range(depth):
        leaf |= (X[:, fi[d]] > fv[d]).astype(np.int64) << (depth - 1 - d)
    vals = np.asarray(e["values"])
    assert np.array_equal(pred, (np.float32(0) - vals[leaf]).astype(np.float32))
    cnt = np.bincount(leaf, minlength=16)
    ew = np.asarray(e["edge_weights"]).astype(np.float64)
    assert np.allclose(np.prod(ew, axis=1) * N, cnt, rtol=1e-5, atol=0.5)
    for l in range(16):
        if cnt[l]:
            want = G[leaf == l].astype(np.float64).mean(axis=0)
            assert np.max(np.abs(vals[l] - want) / np.maximum(np.abs(want), 0.5)) < 1e-6


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_gradients_are_rejected_and_leave_the_model_unchanged(bad):
    import gbrl_amd
    rng = np.random.default_rng(3)
    N, F, D = 5000, 6, 3
    X = rng.standard_normal((N, F)).astype(np.float32)
    G = rng.standard_normal((N, D)).astype(np.float32)
    case = dict(name="nf", seed=0, N=N, F=F, D=D, depth=3, n_bins=64, score="L2", gen="Quantile", policy="greedy", trees=1)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, None, G, None)
    before = {k: np.asarray(v).copy() for k, v in m.get_ensemble_data().items() if k in K.ENSEMBLE_KEYS}
    Gb = G.copy()
    Gb[1234, 1] = bad
    with pytest.raises(RuntimeError):
        m.step(X, None, Gb)
    assert m.get_num_trees() == 1
    after = m.get_ensemble_data()
    for k in K.ENSEMBLE_KEYS:
        assert np.array_equal(before[k], np.asarray(after[k])), k
    m.step(X, None, G)                      # the model keeps working afterwards
    assert m.get_num_trees() == 2


@pytest.mark.parametrize("name", ["obl_l2_q_cat", "grd_cos_u_cat", "grd_l2_q_catonly", "obl_cos_q_cat_rmse"])
def test_device_and_host_categorical_candidates_agree(name, monkeypatch):
    """step() finds the distinct categories of a batch on the device and replays them into the reference's container in the
    reference's insertion order; the fallback scans every cell on the host like the reference.  Same candidates, same order,
    same trees -- and both equal the reference fixture."""
    case, g, (X, Xc, G, y) = load_golden(name)
    outs = []
    for host in ("0", "1"):
        monkeypatch.setenv("GBRL_HIP_HOST_CATEGORICAL", host)
        m, pred = _run_product(case, X, Xc, G, y, "cpu")
        e = m.get_ensemble_data()
        assert_structure_equal(e, g)
        outs.append(({k: np.asarray(e[k]) for k in K.ENSEMBLE_KEYS}, pred))
    for k in K.ENSEMBLE_KEYS:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    assert np.array_equal(outs[0][1], outs[1][1])


def test_more_distinct_categories_than_candidates_falls_back_to_the_reference_ranking():
    """More distinct categories than Fc * n_bins: the reference keeps the Fc * n_bins with the largest mean gradient norm
    (split_candidate_generator.cpp:141-149); the device path declines and the host path reproduces it (checked against the
    oracle restatement)."""
    import gbrl_amd
    import oracle
    rng = np.random.default_rng(5)
    N, Fc, B = 600, 2, 8
    toks = np.array([("t%03d" % i).encode() for i in range(40)], dtype="S128")
    Xc = toks[rng.integers(0, 40, size=(N, Fc))]
    G = rng.standard_normal((N, 2)).astype(np.float32)
    case = dict(name="trunc", seed=0, N=N, F=0, Fc=Fc, D=2, depth=3, n_bins=B, score="L2", gen="Quantile", policy="greedy", trees=2)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    pm = np.asarray(K.drive(m, case, None, Xc, G, None))
    r = oracle.OracleGBRL(**K.ctor_kwargs(case))
    pr = np.asarray(K.drive(r, case, None, Xc, G, None))
    assert_structure_equal(m.get_ensemble_data(), r.get_ensemble_data())
    assert rel_err(pm, pr, float(np.abs(G).mean())) <= TOL


def test_categorical_cells_are_matched_like_strcmp_on_the_device():
    """predict() encodes categorical cells on the device (hash of the bytes before the first NUL, dictionary of the model's
    categories).  Bytes after the first NUL must not matter (the reference compares with strcmp, predictor.cpp:215), unknown
    categories must match nothing."""
    import gbrl_amd
    case, g, (X, Xc, G, y) = load_golden("grd_cos_u_cat")
    m, pred = _run_product(case, X, Xc, G, y, "cpu")
    assert (~np.asarray(m.get_ensemble_data()["is_numerics"])).any()          # the ensemble does use categorical conditions
    raw = np.ascontiguousarray(Xc).view(np.uint8).reshape(Xc.shape[0], Xc.shape[1], 128).copy()
    junk = raw.copy()
    rng = np.random.default_rng(0)
    for i in range(junk.shape[0]):
        for f in range(junk.shape[1]):
            z = int(np.argmax(junk[i, f] == 0))
            junk[i, f, z + 1:] = rng.integers(1, 255, 127 - z, dtype=np.uint8)      # garbage behind the terminator
    Xj = junk.reshape(Xc.shape[0], Xc.shape[1] * 128).view("S128").reshape(Xc.shape)
    assert np.array_equal(np.asarray(m.predict(X, Xj, 0, 0)), pred)
    unknown = np.full(Xc.shape, b"never-seen", dtype="S128")
    pu = np.asarray(m.predict(X, unknown, 0, 0))
    assert not np.array_equal(pu, pred)


@pytest.mark.parametrize("D", [5, 11, 18, 22, 30, 40, 64])
@pytest.mark.parametrize("policy,Fc", [("greedy", 0), ("greedy", 2), ("oblivious", 0), ("oblivious", 2)])
def test_fast_predict_kernels_equal_the_general_kernel(policy, Fc, D, monkeypatch):
    """k_predict_obl (oblivious, numeric) and k_predict_grd (greedy: descent of the tree rebuilt from the leaves' paths,
    numeric and categorical conditions) against the general kernel that walks conditions / leaves like the reference: the same
    fused multiply-adds in the same order, so the outputs must be bitwise equal -- over sub-ranges of trees too."""
    import gbrl_amd
    case = dict(name="pk", seed=77, N=6000, F=9, Fc=Fc, D=D, depth=5, n_bins=64, score="Cosine", gen="Quantile", policy=policy,
                trees=9 if D == 5 else 4,
                opts=[dict(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D - 1),
                      dict(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=D - 1, stop_idx=D)])
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, Xc, G, y)
    outs = {}
    for generic in ("0", "1"):
        monkeypatch.setenv("GBRL_HIP_PREDICT_GENERIC", generic)
        outs[generic] = [np.asarray(m.predict(X, Xc, a, b)) for a, b in (((0, 0), (2, 7), (8, 9), (0, 1)) if D == 5 else ((0, 0), (1, 3), (3, 4)))]
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a, b)
    assert np.abs(outs["0"][0]).max() > 0


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("policy,Fc,F", [("oblivious", 0, 12), ("greedy", 0, 7), ("oblivious", 2, 5), ("greedy", 1, 13)])
def test_second_generation_predict_kernel_at_every_depth(policy, Fc, F, depth, monkeypatch):
    """k_predict_obl2 is compiled for 4, 6 and 8 levels (shallower trees are padded in front with a never-true condition; a
    `max_depth` in between pads the leaf tables to the next size) and for trees-per-worker 1..4: every max_depth from 1 to 8, both
    policies, with and without categorical columns, feature counts that are and are not multiples of 4, ensembles that end inside
    a group, sub-ranges that start and stop inside groups -- against the general kernel, bit for bit, and against the
    first-generation fast kernels."""
    import gbrl_amd
    n_trees = 27
    case = dict(name="pd", seed=300 + depth, N=1500, F=F, Fc=Fc, D=3, depth=depth, n_bins=32, score="L2", gen="Uniform", policy=policy, trees=n_trees,
                loop="rmse", y_cat_weight=0.5)
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, Xc, G, y)
    assert m.get_num_trees() == n_trees
    ranges = ((0, 0), (0, 1), (3, 20), (11, 12), (16, 27), (26, 27))
    outs = {}
    for mode, env in (("gen2", {}), ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"}), ("gen1", {"GBRL_HIP_PREDICT_OBL1": "1"}),
                      ("gen2_k1", {"GBRL_HIP_PREDICT_RG": "1", "GBRL_HIP_PREDICT_TT": "4"}), ("gen2_nb1", {"GBRL_HIP_PREDICT_RG": "2", "GBRL_HIP_PREDICT_NB": "1"})):
        for k in ("GBRL_HIP_PREDICT_GENERIC", "GBRL_HIP_PREDICT_OBL1", "GBRL_HIP_PREDICT_RG", "GBRL_HIP_PREDICT_TT", "GBRL_HIP_PREDICT_NB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[mode] = [np.asarray(m.predict(X, Xc, a, b)) for a, b in ranges]
    for mode in ("gen2", "gen1", "gen2_k1", "gen2_nb1"):
        for r, a, b in zip(ranges, outs[mode], outs["generic"]):
            assert np.array_equal(a, b), (mode, r)
    assert np.abs(outs["gen2"][0]).max() > 0


def test_random_sweep_against_the_oracle_has_no_unexplained_mismatch():
    """60 random configurations (shape, policy, score, generator, bins, depth, min_data_in_leaf, categorical columns, discrete
    columns): bit-identical structure or an explained near-tie (tests/neartie.py), values / predictions within 1e-5."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "parity_sweep.py"), "60", "31000"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_sharded_code_path_on_one_gpu(monkeypatch):
    """The row-sharded path of step() (collective hooks at every exchange point, counting quantiles, every node accumulated)
    run with world_size 1 through torch.distributed/RCCL on the real device pointers: must give the single-GPU tree."""
    import torch
    import torch.distributed as dist
    import gbrl_amd
    from gbrl_amd.dist import install_torch_collective
    case, g, (X, Xc, G, y) = load_golden("obl_l2_q_d6")
    m0, p0 = _run_product(case, X, Xc, G, y, "cpu")
    monkeypatch.setenv("GBRL_HIP_FORCE_COLLECTIVE", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        m1 = gbrl_amd.GBRL(**K.ctor_kwargs(case))
        coll = install_torch_collective(m1, torch.device("cuda:0"))
        p1 = np.asarray(K.drive(m1, case, X, Xc, G, y))
        assert coll.calls > 10 and coll.bytes > 0
    finally:
        dist.destroy_process_group()
    e0, e1 = m0.get_ensemble_data(), m1.get_ensemble_data()
    for k in K.ENSEMBLE_KEYS:
        assert np.array_equal(np.asarray(e0[k]), np.asarray(e1[k])), k
    assert np.array_equal(p0, p1)


def test_native_rccl_exchange_on_one_gpu(monkeypatch):
    """The sharded code path with the model's OWN RCCL communicator (gbrl_hip_set_rccl: all-reduces enqueued on the engine's
    stream, no host synchronisation), world size 1: must give the single-GPU tree."""
    import torch
    import torch.distributed as dist
    import gbrl_amd
    from gbrl_amd.dist import install_rccl
    for name in ("obl_l2_q_d6", "grd_cos_u", "cfg1_rmse_loop"):
        case, g, (X, Xc, G, y) = load_golden(name)
        m0, p0 = _run_product(case, X, Xc, G, y, "cpu")
        monkeypatch.setenv("GBRL_HIP_FORCE_COLLECTIVE", "1")
        monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
        monkeypatch.setenv("MASTER_PORT", "29579")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        try:
            m1 = gbrl_amd.GBRL(**K.ctor_kwargs(case))
            install_rccl(m1, torch.device("cuda:0"))
            p1 = np.asarray(K.drive(m1, case, X, Xc, G, y))
        finally:
            dist.destroy_process_group()
        monkeypatch.delenv("GBRL_HIP_FORCE_COLLECTIVE")
        e0, e1 = m0.get_ensemble_data(), m1.get_ensemble_data()
        for k in K.ENSEMBLE_KEYS:
            assert np.array_equal(np.asarray(e0[k]), np.asarray(e1[k])), (name, k)
        assert np.array_equal(p0, p1)


# ---- inspection of a model grown HERE on the GPU against the reference's fixtures ----
def _tokens(text):
    import re
    return re.findall(r"-?\d+\.?\d*(?:e[-+]?\d+)?|[^\s\d]+|\s+", text)


@pytest.mark.parametrize("name", ["obl_l2_q", "obl_l2_q_cat", "grd_l2_q_mdl", "grd_cos_u_cat"])
def test_inspection_of_a_gpu_grown_model_matches_the_reference(name, tmp_path, capfd):
    """The trees are grown by the HIP path; SHAP values, the exported header and print_tree must then agree with what the
    reference produced for ITS OWN model of the same case: same text token for token, numbers within the value tolerance
    (the trees are bit-identical in structure and 1e-5 close in leaf values; these cases have max_depth 4, where float32
    Linear TreeSHAP is still well conditioned -- at depth 6 the reference's own builds differ by 6% of the array's scale,
    tests/golden/make_explain_golden.py prints it)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "explain_" + name + ".npz"))
    case = K.BY_NAME[name]
    X, Xc, G, y = K.make_inputs(case)
    assert K.inputs_digest(X, Xc, G, y) == str(g["inputs_sha256"])
    m, _ = _run_product(case, X, Xc, G, y, "cpu")
    n = K.EXPLAIN_ROWS
    xs = None if X is None else np.ascontiguousarray(X[:n])
    xcs = None if Xc is None else np.ascontiguousarray(Xc[:n])
    got = m.ensemble_shap(xs, xcs, g["norm_values"], g["base_poly"], g["offset"])
    want = g["shap_ensemble"]
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()

    def same_text(a, b, what):
        ta, tb = _tokens(a), _tokens(b)
        assert len(ta) == len(tb), what
        for x, z in zip(ta, tb):
            if x == z:
                continue
            fx, fz = float(x), float(z)           # raises if a non-numeric token differs
            assert abs(fx - fz) <= 1e-4 * max(abs(fz), 1e-2), (what, x, z)

    capfd.readouterr()
    m.print_tree(0)
    out = capfd.readouterr().out        # other libraries (RCCL's version banner) may write to the same stdout: cut the tree's text out
    start, stop = out.index(" DecisionTree idx: 0"), out.index("******************\n") + 19
    start = out.rindex("\n", 0, start) + 1 if "\n" in out[:start] else 0
    same_text(out[start:stop], g["print_tree_0"].tobytes().decode("latin-1"), "print_tree")
    for k, (mname, fmt, typ, prefix) in enumerate(K.EXPLAIN_CASES[name]):
        if fmt != "float":
            continue
        h = tmp_path / ("m%d.h" % k)
        assert m.export(str(h), mname, fmt, typ, prefix) == 0
        same_text(h.read_text(), g["export_%d" % k].tobytes().decode(), "export")


@pytest.mark.parametrize("name", list(K.EXPLAIN_CASES))
def test_device_shap_equals_the_host_evaluation_and_the_reference(name, tmp_path, monkeypatch):
    """k_shap (one thread per (sample, output), the tree walk as a uniform program) makes the host evaluation's roundings, so the
    two must agree BITWISE; both are within 1e-5 of the reference fixture's scale."""
    import gbrl_amd
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "explain_" + name + ".npz"))
    case = K.BY_NAME[name]
    X, Xc, G, y = K.make_inputs(case)
    p = tmp_path / "ref.gbrl_model"
    p.write_bytes(g["model_file"].tobytes())
    m = gbrl_amd.GBRL.load(str(p))
    n = K.EXPLAIN_ROWS
    xs = None if X is None else np.ascontiguousarray(X[:n])
    xcs = None if Xc is None else np.ascontiguousarray(Xc[:n])
    args = (g["norm_values"], g["base_poly"], g["offset"])
    T = int(g["n_trees"])
    res = {}
    for host in ("0", "1"):
        monkeypatch.setenv("GBRL_HIP_SHAP_HOST", host)
        res[host] = [m.ensemble_shap(xs, xcs, *args)] + [m.tree_shap(t, xs, xcs, *args) for t in sorted({0, T // 2, T - 1})]
    for a, b in zip(res["0"], res["1"]):
        assert np.array_equal(a, b)
    want = [g["shap_ensemble"]] + [g["shap_tree_%d" % t] for t in sorted({0, T // 2, T - 1})]
    for a, w in zip(res["0"], want):
        assert np.abs(a - w).max() <= 1e-5 * np.abs(w).max()


def test_device_shap_at_scale_equals_the_host_evaluation(monkeypatch):
    """A config-2 shaped model in miniature (oblivious, depth 6, 8 outputs, 24 trees) explained on 20 000 rows: device == host bitwise."""
    import time
    import gbrl_amd
    case = dict(name="shap_scale", seed=5, N=20000, F=24, Fc=0, D=8, depth=6, n_bins=64, score="L2", gen="Quantile", policy="oblivious", trees=24)
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, Xc, G, y)
    base, norm, offset = K.poly_vectors(case["depth"])
    out = {}
    for host in ("0", "1"):
        monkeypatch.setenv("GBRL_HIP_SHAP_HOST", host)
        t0 = time.time()
        out[host] = m.ensemble_shap(X, None, norm, base, offset)
        print("ensemble_shap %s: %.3f s" % ("host" if host == "1" else "device", time.time() - t0))
    assert out["0"].shape == (case["N"], case["F"], case["D"]) and np.abs(out["0"]).max() > 0
    assert np.array_equal(out["0"], out["1"])


@pytest.mark.parametrize("policy,Fc", [("greedy", 0), ("oblivious", 0), ("oblivious", 2), ("greedy", 1)])
def test_small_batch_prediction_over_a_large_ensemble_spreads_the_trees(policy, Fc, monkeypatch):
    """An agent acting: a few hundred rows, hundreds of trees.  The fast kernels spread tree ranges over blocks and add the partial
    sums in tree order (kern::predict), so the result is not bitwise the one-chain-per-row sum of the general kernel (which the
    other tests pin against the oracle): it must agree with it to float32 rounding of the sum.  Sub-ranges too."""
    import gbrl_amd
    case = dict(name="act", seed=91, N=700, F=7, Fc=Fc, D=3, depth=3, n_bins=32, score="L2", gen="Quantile", policy=policy, trees=260,
                opts=[dict(algo="SGD", scheduler="Const", init_lr=0.05, start_idx=0, stop_idx=3)])
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    rng = np.random.default_rng(3)
    m.set_feature_weights(np.ones(case["F"] + Fc, np.float32))
    m.set_optimizer("SGD", "Const", 0.05, 0, 3)
    for t in range(case["trees"]):
        m.step(X, Xc, np.ascontiguousarray(np.roll(G, 7 * t, axis=0) + rng.standard_normal(1).astype(np.float32) * 0.1))
    scale = float(np.abs(G).mean())
    for a, b in ((0, 0), (3, 250), (100, 260)):
        monkeypatch.setenv("GBRL_HIP_PREDICT_GENERIC", "0")
        split = np.asarray(m.predict(X, Xc, a, b))
        monkeypatch.setenv("GBRL_HIP_PREDICT_GENERIC", "1")
        chain = np.asarray(m.predict(X, Xc, a, b))
        assert np.abs(chain).max() > 0 and np.abs(split - chain).max() <= 2e-6 * max(np.abs(chain).max(), scale)


@pytest.mark.parametrize("policy,Fc,D", [("oblivious", 0, 5), ("greedy", 0, 5), ("oblivious", 2, 8), ("greedy", 2, 3)])
def test_two_launch_chain_predict_has_the_bits_of_the_general_kernel(policy, Fc, D, monkeypatch):
    """Up to 8192 rows against >= 512 trees go through kern::predict_chain (leaf search spread over the chip, then one fused
    multiply-add chain per (row, output) relayed between the waves of a block).  Same operands, same order, same fused operation as
    the one-thread-per-row general kernel: the same bits -- for both policies, numeric and categorical conditions, outputs without an
    optimizer, batches of 1 .. 3000 rows, the whole range and sub-ranges that start / end inside a 64-tree batch; forced
    (GBRL_HIP_PREDICT_CHAIN=1) for the short ranges the dispatcher would not give it."""
    import gbrl_amd
    rng = np.random.default_rng(3)
    F, N, T = 7, 3000, 700
    X = rng.standard_normal((N, F), dtype=np.float32)
    Xc = K.TOKENS[rng.integers(0, 6, size=(N, Fc))] if Fc else None
    m = gbrl_amd.GBRL(input_dim=F + Fc, output_dim=D, policy_dim=D, max_depth=5, min_data_in_leaf=0, n_bins=64, par_th=10, cv_beta=0.9,
                      split_score_func="cosine", generator_type="Quantile", use_control_variates=False, batch_size=5000, grow_policy=policy,
                      verbose=0, device="cpu", learner_name="chain")
    m.set_bias(rng.standard_normal(D).astype(np.float32)); m.set_feature_weights(np.ones(F + Fc, np.float32))
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.05, start_idx=0, stop_idx=D - 2)
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.01, start_idx=D - 2, stop_idx=D - 1)     # the last output has no optimizer
    m.set_feature_mapping(np.arange(F + Fc, dtype=np.int32), np.array([True] * F + [False] * Fc))
    for t in range(T):
        rows = rng.integers(0, N, size=512)
        G = rng.standard_normal((512, D), dtype=np.float32) + np.float32(0.5) * X[rows][:, :1]
        if Fc:
            G[:, 0] += (Xc[rows][:, 0] == K.TOKENS[1]).astype(np.float32) * 2
        m.step(np.ascontiguousarray(X[rows]), None if Xc is None else np.ascontiguousarray(Xc[rows]), G)
    assert m.get_num_trees() == T
    for n in (1, 5, 64, 257, 3000):
        xa, xc = np.ascontiguousarray(X[:n]), (None if Xc is None else np.ascontiguousarray(Xc[:n]))
        for a, b in ((0, 0), (0, 1), (3, 40), (100, 700), (0, 63), (0, 64), (0, 65), (5, 133), (60, 700), (0, 512)):
            monkeypatch.setenv("GBRL_HIP_PREDICT_CHAIN", "0"); monkeypatch.setenv("GBRL_HIP_PREDICT_GENERIC", "1")
            want = np.asarray(m.predict(xa, xc, a, b))
            monkeypatch.delenv("GBRL_HIP_PREDICT_GENERIC")
            monkeypatch.setenv("GBRL_HIP_PREDICT_CHAIN", "1")
            got = np.asarray(m.predict(xa, xc, a, b))
            assert np.array_equal(got, want), (n, a, b)
            monkeypatch.delenv("GBRL_HIP_PREDICT_CHAIN")
            if (b if b else T) - a >= 512 and n >= 640:      # the dispatcher's own choice for these shapes (fewer rows: thread slices)
                assert np.array_equal(np.asarray(m.predict(xa, xc, a, b)), want), (n, a, b)


def test_fit_past_512_trees_uses_the_chain_path_and_grows_the_same_model(monkeypatch):
    """fit() predicts trees [0, i) for every batch; from 512 trees on those predictions go through kern::predict_chain, which has
    the bits of the one-chain-per-row kernel: the 560-tree model and the returned loss are identical with the path switched off."""
    import gbrl_amd
    rng = np.random.default_rng(21)
    N, F, D = 700, 5, 2
    X = rng.standard_normal((N, F), dtype=np.float32)
    y = (np.tanh(X[:, :D]) + 0.2 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
    out = []
    for chain in ("1", "0"):
        monkeypatch.setenv("GBRL_HIP_PREDICT_CHAIN", chain) if chain == "0" else monkeypatch.delenv("GBRL_HIP_PREDICT_CHAIN", raising=False)
        m = gbrl_amd.GBRL(input_dim=F, output_dim=D, policy_dim=D, max_depth=3, min_data_in_leaf=0, n_bins=32, par_th=10, cv_beta=0.9,
                          split_score_func="L2", generator_type="Quantile", use_control_variates=False, batch_size=256, grow_policy="oblivious",
                          verbose=0, device="cpu", learner_name="fitchain")
        m.set_feature_weights(np.ones(F, np.float32))
        m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.05, start_idx=0, stop_idx=D)
        m.set_feature_mapping(np.arange(F, dtype=np.int32), np.ones(F, dtype=bool))
        loss = m.fit(X, None, y, 560, False, "MultiRMSE")
        if chain == "0":
            monkeypatch.setenv("GBRL_HIP_PREDICT_GENERIC", "1")      # the one-thread-per-row chain
        out.append((loss, m.get_ensemble_data(), np.asarray(m.predict(X, None, 0, 0))))
        monkeypatch.delenv("GBRL_HIP_PREDICT_GENERIC", raising=False)
    (l1, e1, p1), (l0, e0, p0) = out
    assert int(np.asarray(e1["depths"]).size) == 560
    assert l1 == l0
    for k in K.ENSEMBLE_KEYS:
        assert np.array_equal(np.asarray(e1[k]), np.asarray(e0[k])), k
    assert np.array_equal(p1, p0)
