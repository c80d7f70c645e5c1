"""CPU-only sweep (authoring container or GPU box): the oracle RESTATEMENT (oracle/oracle.cpp) against the REAL reference build
(oracle/_ref) on random cases, including regimes no committed fixture covers (many outputs, many bins, weights with zeros, many
categorical columns).  Structure must be bit-identical or an explained near-tie; values / predictions within 1e-5.
    OMP_NUM_THREADS=8 python scripts/oracle_vs_ref_sweep.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import neartie
from helpers import assert_structure_equal, assert_values_close, rel_err
import oracle

ref = oracle.load_ref()
assert ref is not None
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(seed0)
exact = near = bad = 0
t0 = time.time()
for i in range(n_cases):
    Fc = int(rng.choice([0, 0, 1, 3, 8]))
    case = dict(name="ovr%d" % i, seed=seed0 + i, N=int(rng.choice([300, 900, 2000])), F=int(rng.choice([0 if Fc else 1, 2, 7, 20])), Fc=Fc,
                D=int(rng.choice([1, 3, 8, 13, 18, 33])), depth=int(rng.choice([1, 3, 4, 5])), n_bins=int(rng.choice([7, 32, 256, 500])),
                score=str(rng.choice(["L2", "Cosine"])), gen=str(rng.choice(["Quantile", "Uniform"])), policy=str(rng.choice(["greedy", "oblivious"])),
                trees=1, min_data_in_leaf=int(rng.choice([0, 0, 7])), n_tokens=int(rng.choice([3, 8, 20])))
    if case["F"] + case["Fc"] == 0: case["F"] = 2
    if case["N"] < case["n_bins"] + 1: case["n_bins"] = 32
    if rng.random() < 0.5:
        nin = case["F"] + case["Fc"]
        case["feature_weights"] = [float(v) for v in rng.choice([0.0, 0.5, 1.0, 1.0, 2.0], nin)]
        if max(case["feature_weights"]) == 0.0: case["feature_weights"][0] = 1.0
    X, Xc, G, y = K.make_inputs(case)
    o = oracle.OracleGBRL(**K.ctor_kwargs(case)); po = np.asarray(K.drive(o, case, X, Xc, G, y))
    r = ref.GBRL(**K.ctor_kwargs(case)); pr = np.asarray(K.drive(r, case, X, Xc, G, y))
    eo, er = o.get_ensemble_data(), r.get_ensemble_data()
    scale = float(np.abs(G).mean())
    try:
        assert_structure_equal(eo, er); assert_values_close(eo, er, scale, 1e-5); assert rel_err(po, pr, scale) <= 1e-5
        exact += 1
    except AssertionError as ex:
        info = neartie.explain_first_mismatch(case, X, Xc, G, er, eo)
        if info and info.get("explained"): near += 1
        else:
            bad += 1; print("MISMATCH", case, str(ex)[:160], info, flush=True)
print("oracle vs reference: cases %d, exact %d, explained near-ties %d, unexplained %d  (%.1f s)" % (n_cases, exact, near, bad, time.time() - t0))
sys.exit(1 if bad else 0)
