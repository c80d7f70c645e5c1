"""Persistent / resident mode of the second-generation oblivious predict kernel at max_depth 7 and 8 (ADVICE r03 "high").

`k_predict_obl2<..., PERSISTENT>` keeps 256 x (blocks per CU) blocks resident, each walking row tiles, when the batch holds at least two
tiles per resident block (n >= 65 536 rows here); with at most 16 trees whose leaf values fit the two 8-tree value buffers it also stages
the whole ensemble's values ONCE per block ("resident values").  The staging loop moves at most 16 KiB per 8-tree group: depth 7-8 trees
with 3-4 outputs (4 KiB of values per tree) exceed it, and the round-3 gate let them through (out-of-bounds LDS staging); the gate in
`predict_obl2.hip` (8 * bytes per tree <= 16 KiB) now sends them to the regular plan.  No test reached that branch: the depth sweep of
test_gpu_parity.py uses 1500 rows (below the persistent threshold) and the register-tile sweep stops at depth 6.

Here: oblivious depth 7 and 8, 1 .. 4 outputs, 5 / 8 / 16 trees, 65 536+ rows, 16 / 64 / 128 features -- the default dispatch against
`GBRL_HIP_PREDICT_NO_RESIDENT=1` (persistent blocks, values staged per group), `GBRL_HIP_PREDICT_NO_PERSIST=1` (one block per tile) and
`GBRL_HIP_PREDICT_GENERIC=1` (the general kernel), bit for bit.  (The hooks are read per call.)
"""
import numpy as np
import pytest

import cases as K

pytestmark = pytest.mark.gpu

HOOKS = ("GBRL_HIP_PREDICT_NO_RESIDENT", "GBRL_HIP_PREDICT_NO_PERSIST", "GBRL_HIP_PREDICT_GENERIC")


def _model(depth, D, F, trees, seed):
    import gbrl_amd
    case = dict(name="deep", seed=seed, N=3000, F=F, Fc=0, D=D, depth=depth, n_bins=64, score="L2", gen="Uniform", policy="oblivious", trees=trees)
    X, Xc, G, y = K.make_inputs(case)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case))
    K.drive(m, case, X, Xc, G, y)
    assert m.get_num_trees() == trees and int(np.asarray(m.get_ensemble_data()["depths"]).max()) == depth
    return m


@pytest.mark.parametrize("depth", [7, 8])
@pytest.mark.parametrize("D", [1, 2, 3, 4])
@pytest.mark.parametrize("trees,F", [(5, 16), (8, 64), (16, 128), (16, 16)])
def test_persistent_and_resident_predict_at_depth_7_and_8(depth, D, trees, F, monkeypatch):
    m = _model(depth, D, F, trees, seed=7000 + 10 * depth + D)
    rng = np.random.default_rng(depth * 100 + D * 10 + trees)
    n = 65536 + 64 * int(rng.integers(0, 40)) + int(rng.integers(0, 64))      # full tiles and a ragged last one
    X = rng.standard_normal((n, F)).astype(np.float32)
    outs = {}
    for mode, env in (("default", {}), ("no_resident", {"GBRL_HIP_PREDICT_NO_RESIDENT": "1"}), ("no_persist", {"GBRL_HIP_PREDICT_NO_PERSIST": "1"}),
                      ("generic", {"GBRL_HIP_PREDICT_GENERIC": "1"})):
        for k in HOOKS:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[mode] = [np.asarray(m.predict(X, None, a, b)) for a, b in ((0, 0), (1, trees), (0, min(trees, 8)))]
    for mode in ("default", "no_resident", "no_persist"):
        for a, b in zip(outs[mode], outs["generic"]):
            assert a.shape == b.shape and a.tobytes() == b.tobytes(), mode
    assert np.abs(outs["default"][0]).max() > 0
