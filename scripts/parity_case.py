"""Reproduce one sweep case and print both trees:  python scripts/parity_case.py '<case dict repr>' """
import os, sys, ast
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K, neartie
import gbrl_amd, oracle
case = ast.literal_eval(sys.argv[1])
case["trees"] = int(sys.argv[2]) if len(sys.argv) > 2 else case["trees"]
X, Xc, G, y = K.make_inputs(case)
m = gbrl_amd.GBRL(**K.ctor_kwargs(case)); K.drive(m, case, X, Xc, G, y)
r = oracle.OracleGBRL(**K.ctor_kwargs(case)); K.drive(r, case, X, Xc, G, y)
e, o = m.get_ensemble_data(), r.get_ensemble_data()
np.set_printoptions(linewidth=200, precision=6)
for k in ("tree_indices", "depths", "feature_indices", "feature_values", "inequality_directions", "edge_weights", "values"):
    a, b = np.asarray(e[k]), np.asarray(o[k])
    print("==", k, "equal" if a.shape == b.shape and np.array_equal(a, b) else "DIFF")
    if not (a.shape == b.shape and np.array_equal(a, b)):
        print("product:\n", a[:12]); print("oracle:\n", b[:12])
print("distinct values in col 0:", np.unique(X[:, 0]).size if X is not None and X.shape[1] else None)
if case["trees"] == 1:
    print(neartie.explain_first_mismatch(case, X, Xc, G, o, e))
