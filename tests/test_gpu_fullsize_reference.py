"""The headline configuration against the REFERENCE ITSELF (north_star's acceptance sentence): the product's tree at
batch = 2^20, n_feat = 128, depth = 6, D = 8 compared with the tree the reference's own CPU path (oracle/_ref, built from
/root/reference by oracle/Makefile) grew on the same inputs in the build container -- tests/golden/full_cfg2.npz (BASELINE configs[1]:
oblivious / L2 / quantile; full_cfg2_s1: a second seed with a four times weaker signal) and tests/golden/full_cfg3.npz (configs[2]: greedy /
Cosine / policy + value optimisers; capacity-patched build; full_cfg3_s1: second seed, weaker signal), made by
tests/golden/make_fullsize_golden.py (835 - 1230 s / tree on 8 vCPU).  The 512 MiB of inputs are regenerated from the
seed (integer PCG64 draws + exactly rounded float32 arithmetic only) and checked through their SHA-256 before anything is compared.

Bar: bit-identical structure.  Two modes:
  * GBRL_HIP_NEARTIE_MAX_ROWS=0 -- every flagged node re-scored in the reference's float32 operation sequence (serial chains over up to 2^20
    rows; evaluated in parallel by seqsum.hip: 14-22 ms per tree): configs[1] 6 of 6 levels, configs[2] 63 of 63 internal nodes, on both seeds of
    both, asserted exactly;
  * default -- batches above 65 536 rows keep the exact float64 arg-max (1.9 ms per tree): configs[1] 6 of 6 (both seeds); configs[2] 22 of 63 (28 of 63 on the second seed) -- the
    reference's float32 noise (4e-5 relative at a 522 256-row node) prefers a neighbouring threshold whose true score is 5e-6 lower, and the
    41 nodes below it sit on another partition.  Where a level / node differs, the float64 scores of both candidates on the node's rows are
    printed next to the reference's own float32 summation noise (eps32 * sqrt(rows), node.cpp:336-352 sums sequentially in float32), and
    the COUNT of differing levels / nodes is asserted against the committed bounds below.
"""
import json
import os

import numpy as np
import pytest

import cases as K
import fullsize

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
# committed bounds (measured on MI355X, round 6; see README "Parity at the headline size")
MAX_DIFFERING_LEVELS_CFG2 = {"replay": 0, "default": 0}
MAX_DIFFERING_NODES_CFG3 = {"full_cfg3": {"replay": 0, "default": 41}, "full_cfg3_s1": {"replay": 0, "default": 35}}
MODES = {"replay": "0", "default": None}     # GBRL_HIP_NEARTIE_MAX_ROWS


def _fixture(name):
    path = os.path.join(HERE, "golden", name + ".npz")
    if not os.path.exists(path):
        pytest.skip(name + ".npz not generated (tests/golden/make_fullsize_golden.py, build container only)")
    fx = np.load(path)
    case = json.loads(str(fx["case_json"]))
    X, Xc, G, y = K.make_inputs(case)
    assert K.inputs_digest(X, Xc, G, y) == str(fx["inputs_sha256"]), "regenerated inputs differ from the ones the reference saw"
    return fx, case, X, G


def _grow(case, X, G, monkeypatch, root_mode=None, mode="default"):
    import gbrl_amd
    monkeypatch.delenv("GBRL_HIP_NO_NEARTIE_REPLAY", raising=False)
    if MODES[mode] is None:
        monkeypatch.delenv("GBRL_HIP_NEARTIE_MAX_ROWS", raising=False)
    else:
        monkeypatch.setenv("GBRL_HIP_NEARTIE_MAX_ROWS", MODES[mode])
    if root_mode is None:
        monkeypatch.delenv("GBRL_HIP_ROOT_COUNTS", raising=False)
    else:
        monkeypatch.setenv("GBRL_HIP_ROOT_COUNTS", root_mode)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, None, G, None)
    return {k: np.asarray(v) for k, v in m.get_ensemble_data().items()}


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("mode", ["replay", "default"])
@pytest.mark.parametrize("fixture", ["full_cfg2", "full_cfg2_s1"])
def test_config2_tree_is_the_reference_tree_at_full_size(fixture, mode, monkeypatch):
    fx, case, X, G = _fixture(fixture)
    e = _grow(case, X, G, monkeypatch, mode=mode)
    depth_ref, depth = int(fx["depths"][0]), int(e["depths"][0])
    fi_r, fv_r, fi, fv = fx["feature_indices"][0], fx["feature_values"][0], e["feature_indices"][0], e["feature_values"][0]
    same = [bool(l < depth and fi[l] == fi_r[l] and _bits(fv[l:l + 1])[0] == _bits(fv_r[l:l + 1])[0]) for l in range(depth_ref)]
    lead = next((l for l, s in enumerate(same) if not s), depth_ref)
    print("config 2 (%s, %s) at 2^20 x 128: %d of %d levels identical to the reference's tree (%d leading); reference %s / product %s"
          % (fixture, mode, sum(same), depth_ref, lead, list(zip(fi_r.tolist(), fv_r.tolist()))[:depth_ref], list(zip(fi.tolist(), fv.tolist()))[:depth]))
    if lead < depth_ref:
        # the first differing level: float64 scores of both choices on the partition of the common prefix, against the reference's noise
        thr = fullsize.quantile_thresholds(X, case["n_bins"])
        node = np.zeros(X.shape[0], np.int64)
        for l in range(lead):
            node = node * 2 + (X[:, fi_r[l]] > fv_r[l])
        bg = fullsize.standardise(G)
        feats = sorted({int(fi_r[lead]), int(fi[lead])})
        codes = fullsize.class_codes(X[:, feats], thr[feats])
        sc = fullsize.candidate_scores(codes, node, 1 << lead, bg, case["n_bins"], False).sum(axis=0)
        def score_of(f, v):
            b = int(np.nonzero(_bits(thr[f]) == _bits(np.array([v], np.float32))[0])[0][0])
            return float(sc[feats.index(int(f)), b]), b
        s_ref, b_ref = score_of(fi_r[lead], fv_r[lead])
        s_got, b_got = score_of(fi[lead], fv[lead])
        print("  level %d: reference (%d, bin %d) float64 score %.9g | product (%d, bin %d) %.9g | gap %.2e relative; reference float32 noise eps32*sqrt(N) = %.1e"
              % (lead, fi_r[lead], b_ref, s_ref, fi[lead], b_got, s_got, (s_got - s_ref) / abs(s_got), 2.0 ** -24 * np.sqrt(X.shape[0])))
    assert depth == depth_ref
    assert depth_ref - sum(same) <= MAX_DIFFERING_LEVELS_CFG2[mode], same
    if all(same):
        assert np.array_equal(e["tree_indices"], fx["tree_indices"]) and np.array_equal(e["inequality_directions"], fx["inequality_directions"])
        scale = float(np.abs(G).mean())
        err = float(np.max(np.abs(e["values"] - fx["values"]) / np.maximum(np.abs(fx["values"]), scale)))
        werr = float(np.max(np.abs(e["edge_weights"] - fx["edge_weights"])))
        print("  leaf values: max relative error %.2e (bar 1e-5); edge weights max abs difference %.1e" % (err, werr))
        assert err <= 1e-5 and werr <= 1e-6
    # the cross-check mode of the root level (class counts from the selection's ranks AND accumulated, compared entry by entry inside the engine)
    if mode == "default" and fixture == "full_cfg2":
        e2 = _grow(case, X, G, monkeypatch, root_mode="2")
        for k in e:
            assert np.asarray(e[k]).tobytes() == np.asarray(e2[k]).tobytes(), k


def _greedy_nodes(e):
    """{path prefix -> split} of a greedy tree stored leaf by leaf (types.h:279-304): the prefix is the sequence of
    (feature, threshold bits, direction) decisions above the node."""
    fi, fvb, dirs, dep = np.asarray(e["feature_indices"]), _bits(np.asarray(e["feature_values"])).reshape(np.asarray(e["feature_values"]).shape), \
        np.asarray(e["inequality_directions"]), np.asarray(e["depths"])
    nodes = {}
    for leaf in range(len(dep)):
        prefix = ()
        for k in range(int(dep[leaf])):
            split = (int(fi[leaf, k]), int(fvb[leaf, k]))
            assert nodes.setdefault(prefix, split) == split
            prefix = prefix + ((split[0], split[1], int(dirs[leaf, k])),)
    return nodes


@pytest.mark.parametrize("mode", ["replay", "default"])
@pytest.mark.parametrize("fixture", ["full_cfg3", "full_cfg3_s1"])
def test_config3_tree_is_the_reference_tree_at_full_size(fixture, mode, monkeypatch):
    fx, case, X, G = _fixture(fixture)
    e = _grow(case, X, G, monkeypatch, mode=mode)
    ref = {k: fx[k] for k in fx.files}
    n_ref, n_got = _greedy_nodes(ref), _greedy_nodes(e)
    same = sum(1 for p, s in n_ref.items() if n_got.get(p) == s)
    print("config 3 (%s, %s) at 2^20 x 128: %d of the reference's %d internal nodes identical (product grew %d); leaves %d / %d"
          % (fixture, mode, same, len(n_ref), len(n_got), len(ref["depths"]), len(e["depths"])))
    differing = [(p, s, n_got.get(p)) for p, s in n_ref.items() if n_got.get(p) != s]
    top = sorted(differing, key=lambda t: len(t[0]))[:4]
    first_gap = None
    if top:
        thr = fullsize.quantile_thresholds(X, case["n_bins"])
        for p, s, g in top:
            rows = np.ones(X.shape[0], bool)
            for (f, vb, d) in p:
                rows &= (X[:, f] > np.array([vb], np.uint32).view(np.float32)[0]) == bool(d)
            idx = np.nonzero(rows)[0]
            line = "  node at depth %d (%d rows): reference split %s, product %s" % (len(p), len(idx), s, g)
            if g is not None:
                feats = sorted({s[0], g[0]})
                codes = fullsize.class_codes(X[idx][:, feats], thr[feats])
                sc = fullsize.candidate_scores(codes, np.zeros(len(idx), np.int64), 1, np.asarray(G, np.float64)[idx], case["n_bins"], True)[0]
                def score_of(f, vb):
                    return float(sc[feats.index(f), int(np.nonzero(_bits(thr[f]) == vb)[0][0])])
                a, b = score_of(*s), score_of(*g)
                if first_gap is None:
                    first_gap = (a, b, len(idx))
                line += "; float64 scores %.9g / %.9g, gap %.2e relative, reference float32 noise %.1e" % (a, b, (b - a) / abs(b), 2.0 ** -24 * np.sqrt(len(idx)))
            print(line)
    assert len(n_ref) - same <= MAX_DIFFERING_NODES_CFG3[fixture][mode]
    if mode == "default" and first_gap is not None:
        # the shallowest difference must be an explained near-tie: the product at (or above) the reference's true score, the gap inside its noise
        a, b, n_rows = first_gap
        assert b >= a and (b - a) / abs(b) <= 4 * 2.0 ** -24 * np.sqrt(n_rows) + 1e-12, first_gap
    if same == len(n_ref) == len(n_got):
        for k in ("depths", "feature_indices", "inequality_directions", "tree_indices"):
            assert np.array_equal(e[k], ref[k]), k
        assert np.array_equal(_bits(e["feature_values"]), _bits(ref["feature_values"]))
        scale = float(np.abs(G).mean())
        err = float(np.max(np.abs(e["values"] - ref["values"]) / np.maximum(np.abs(ref["values"]), scale)))
        print("  leaf values: max relative error %.2e (bar 1e-5)" % err)
        assert err <= 1e-5
