"""A 4-category column split three times by one oblivious tree: the last two categories give the SAME partition, an exact tie.  The product
takes the lowest candidate index, the reference whatever the rounding of its contracted float32 score expression prefers -- the trees differ in
that one categorical value (an "explained near-tie").  Kept as the worked example behind the cardinalities chosen in
tests/test_gpu_edges.py::test_categorical_steps_with_changing_cardinality_and_weights."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import cases as K
import gbrl_amd, oracle

def run(variant):
    F, Fc, D = 3, 4, 2
    case = dict(name="catjump", seed=0, N=2000, F=F, Fc=Fc, D=D, depth=4, n_bins=256, score="L2", gen="Quantile", policy="oblivious", trees=0)
    X, Xc, G, y = K.make_inputs(case)
    rng = np.random.default_rng(11)
    N = X.shape[0]
    def cells(card):
        out = np.empty((N, Fc), dtype="S128")
        for c in range(Fc):
            ids = rng.integers(0, card[c], size=N)
            out[:, c] = np.array([("k%d_%d" % (c, i)).encode() for i in range(card[c])], dtype="S128")[ids]
        return out
    few, many, mid = cells([3, 4, 3, 5]), cells([300, 4, 3, 5]), cells([40, 4, 120, 5])
    g2 = (G + (many[:, 0] == b"k0_7").astype(np.float32)[:, None] * 2.0).astype(np.float32)
    w = np.array([1.0, 0.5, 2.0, 1.0, 3.0, 0.25, 1.0], np.float32)
    steps = [("step", X, few, G), ("step", X, few, G), ("step", X, many, g2), ("step", X[:700], many[:700], g2[:700]), ("w", w),
             ("step", X, mid, g2), ("step", X, few, G), ("step", X, many, g2)]
    if variant == "no_w": steps = [s for s in steps if s[0] != "w"]
    if variant == "no_small": steps = [s for s in steps if not (s[0] == "step" and s[1].shape[0] == 700)]
    if variant == "w_first": steps = [("w", w)] + [s for s in steps if s[0] != "w"]
    models = [gbrl_amd.GBRL(**K.ctor_kwargs(case)), oracle.OracleGBRL(**K.ctor_kwargs(case))]
    for m in models:
        K.drive(m, case, X, Xc, G, y)
        for op in steps:
            if op[0] == "w": m.set_feature_weights(op[1])
            else: m.step(np.ascontiguousarray(op[1]), np.ascontiguousarray(op[2]), np.ascontiguousarray(op[3]))
    e, o = models[0].get_ensemble_data(), models[1].get_ensemble_data()
    T = len(np.asarray(e["depths"]))
    for t in range(T):
        same = all(np.array_equal(np.asarray(e[k])[t], np.asarray(o[k])[t]) for k in ("feature_indices", "is_numerics", "categorical_values", "depths"))
        fv = np.array_equal(np.asarray(e["feature_values"])[t].view(np.uint32), np.asarray(o["feature_values"])[t].view(np.uint32))
        if not (same and fv):
            print(variant, "tree", t, "DIFFERS")
            for k in ("feature_indices", "is_numerics", "feature_values", "categorical_values"):
                print("   ", k, "product", [x if not isinstance(x, bytes) else x.decode() for x in np.asarray(e[k])[t].tolist()], "oracle", [x if not isinstance(x, bytes) else x.decode() for x in np.asarray(o[k])[t].tolist()])
    print(variant, "done,", T, "trees")

for v in ("full", "no_w", "no_small", "w_first"):
    run(v)
