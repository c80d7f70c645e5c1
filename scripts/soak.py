"""Soak: ONE product model takes many steps on batches of changing size (buffer growth / reuse), numeric + categorical; every
new tree is compared with the tree the oracle restatement fits to the same batch and gradients (bit-identical structure or an
explained near-tie), and the model's predictions are re-derived from its own ensemble on the host at the end.
    python scripts/soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import neartie
from helpers import assert_structure_equal, STRUCTURE_KEYS
import gbrl_amd, oracle
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(123)
F, Fc, D = 7, 2, 3
toks = np.array([("c%02d" % i).encode() for i in range(9)], dtype="S128")


def setup(m):
    m.set_feature_weights(np.ones(F + Fc, np.float32))
    m.set_optimizer(algo="SGD", scheduler="Const", init_lr=0.1, start_idx=0, stop_idx=D)
    m.set_feature_mapping(np.arange(F + Fc, dtype=np.int32), np.array([True] * F + [False] * Fc))


def last_tree(e, policy):
    t0 = int(e["tree_indices"][-1])
    out = {}
    for k in K.ENSEMBLE_KEYS:
        a = np.asarray(e[k])
        if k == "tree_indices":
            out[k] = np.zeros(1, a.dtype)
        elif policy == "oblivious" and k in ("depths", "feature_indices", "feature_values", "is_numerics", "categorical_values"):
            out[k] = a[-1:]
        else:
            out[k] = a[t0:]
    return out


for policy in ("greedy", "oblivious"):
    case = dict(name="soak", seed=0, N=0, F=F, Fc=Fc, D=D, depth=4, n_bins=64, score="Cosine" if policy == "greedy" else "L2",
                gen="Quantile", policy=policy, trees=1, min_data_in_leaf=3)
    m = gbrl_amd.GBRL(**K.ctor_kwargs(case)); setup(m)
    exact = near = 0
    t0 = time.time()
    for it in range(steps):
        N = int(rng.choice([70, 300, 1500, 5000, 12000, 333]))
        X = rng.standard_normal((N, F)).astype(np.float32)
        Xc = toks[rng.integers(0, 9, size=(N, Fc))]
        y = (np.sin(X[:, :D]) + (Xc[:, :1] == toks[0]) * 0.7).astype(np.float32)
        G = (np.asarray(m.predict(X, Xc, 0, 0)).reshape(N, D) - y).astype(np.float32)
        m.step(X, Xc, G)
        r = oracle.OracleGBRL(**K.ctor_kwargs(case)); setup(r); r.step(X, Xc, G)
        mine, ref = last_tree(m.get_ensemble_data(), policy), r.get_ensemble_data()
        try:
            assert_structure_equal(mine, ref)
            exact += 1
        except AssertionError as ex:
            info = neartie.explain_first_mismatch(dict(case, N=N), X, Xc, G, ref, mine)
            assert info and info.get("explained"), (it, N, str(ex)[:100], info)
            near += 1
    print("%s: %d steps on one model (%d trees): %d trees bit-identical to the oracle's, %d explained near-ties, %.1f s" % (
        policy, steps, m.get_num_trees(), exact, near, time.time() - t0), flush=True)
