"""Which seeds of the greedy depth-6 fixture (BASELINE configs[2] in miniature) are free of near-ties?  Runs the capacity-patched
reference build and the product side by side (GPU box; oracle/_ref/capacity travels with the snapshot)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
import cases as K, oracle, neartie, gbrl_amd
cap = oracle.load_ref_capacity()
base = K.BY_NAME["grd_cos_q_ac_d6"]
for N in (3000, 6000):
    for seed in range(26, 40):
        case = dict(base, seed=seed, N=N)
        X, Xc, G, y = K.make_inputs(case)
        r = cap.GBRL(**K.ctor_kwargs(case)); K.drive(r, case, X, Xc, G, y)
        m = gbrl_amd.GBRL(**K.ctor_kwargs(case)); K.drive(m, case, X, Xc, G, y)
        g = {k: np.asarray(r.get_ensemble_data()[k]) for k in K.ENSEMBLE_KEYS}
        e = {k: np.asarray(m.get_ensemble_data()[k]) for k in K.ENSEMBLE_KEYS}
        mm = neartie.first_mismatch(g, e, case["policy"])
        if mm is None:
            print("N %d seed %d: identical (leaves %d)" % (N, seed, g["values"].shape[0]), flush=True)
        else:
            try:
                info = neartie.explain_first_mismatch(case, X, Xc, G, g, e)
            except Exception as ex:
                info = repr(ex)
            print("N %d seed %d: mismatch %s -> %s" % (N, seed, mm, info), flush=True)
